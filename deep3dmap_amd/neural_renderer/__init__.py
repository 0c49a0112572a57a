"""MI355X-native mirror of the `neural_renderer` package surface deep3dmap imports
(pnpmodules/neural_renderer/neural_renderer/__init__.py:1-12)."""
from .cameras import get_points_from_angles, look, look_at, perspective, projection
from .mesh_ops import lighting, vertices_to_faces
from .obj_io import Mesh, load_obj, save_obj
from .rasterize import (Rasterize, RasterizeFunction, rasterize, rasterize_depth, rasterize_rgbad,
                        rasterize_silhouettes)
from .renderer import Renderer

__version__ = '1.1.3'
name = 'neural_renderer_pytorch'
