"""Debug: forward binning statistics of the gan2shape workload (reads the forward workspace back)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from deep3dmap_amd import _lib
from deep3dmap_amd.core import NrRenderer
from deep3dmap_amd.neural_renderer import rasterize_ops as ops
b, hw = 16, 64
r = NrRenderer(dict(min_depth=0.9, max_depth=1.1, rot_center_depth=1.0, fov=10, tex_cube_size=2), hw)
rng = np.random.default_rng(0)
noise = torch.from_numpy(rng.standard_normal((b, hw, hw)).astype(np.float32))
depth = (1.0 + 0.1 * torch.tanh(torch.nn.functional.avg_pool2d(noise[:, None], 5, 1, 2)[:, 0])).cuda()
view = (torch.from_numpy(rng.uniform(-1, 1, (b, 6)).astype(np.float32)) * torch.tensor([0.5, 1.0, 0.3, 0.1, 0.1, 0.02])).cuda()
r.set_transform_matrices(view)
w = r.warp_canon_depth(depth)
torch.cuda.synchronize()
ws = ops._workspaces[("fwd", depth.device)].cpu().numpy()
B, S, F = 16, 128, 2 * 7938
T = (S // 8) ** 2
al = lambda v: (v + 255) // 256 * 256
cnt = ws[:B * T * 4].view(np.int32)
big = ws[al(B * T * 4) * 2: al(B * T * 4) * 2 + B * 4].view(np.int32)
print("tile counts: mean %.1f max %d nonzero %d of %d" % (cnt.mean(), cnt.max(), (cnt > 0).sum(), cnt.size))
print("big faces per view:", big.tolist())
