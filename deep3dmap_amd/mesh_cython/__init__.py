"""MI355X-native stand-in for deep3dmap/core/renderer/renderer_demo/mesh_cython (the face3d utility rasterizer
family): `from deep3dmap_amd.mesh_cython import render` gives the module the reference's callers import."""
from . import render  # noqa: F401
