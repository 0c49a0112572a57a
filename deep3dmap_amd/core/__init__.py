"""Mirror of the deep3dmap modules that sit directly on the rasterization path:
deep3dmap/core/renderer/renderer_nr.py (NrRenderer), deep3dmap/core/renderer/renderer_pt3d.py (Pt3dRenderer),
deep3dmap/core/renderer/utils.py (their helpers) and the losses of deep3dmap/core/utils/utils.py."""
from .losses import multiview_fit_loss, photometric_loss, silhouette_loss, smooth_loss
from .renderer_nr import NrRenderer
from .renderer_pt3d import Pt3dRenderer
from .renderer_utils import (get_face_idx, get_grid, get_rotation_matrix, get_textures_from_im,
                             get_transform_matrices, vcolor_to_texture_cube)
