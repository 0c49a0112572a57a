"""deep3dmap_amd -- MI355X-native differentiable mesh rasterization core for deep3dmap.

The package holds only the rasterization hot path: the HIP kernels and their C ABI
(csrc/, include/d3m_raster.h), and the Python mirror of the reference interfaces that sit on that
path (`deep3dmap_amd.neural_renderer` for pnpmodules/neural_renderer, `deep3dmap_amd.core` for
deep3dmap/core/renderer/renderer_nr.py and the losses of deep3dmap/core/utils/utils.py)."""
__version__ = "0.1.0"
