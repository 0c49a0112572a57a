import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep3dmap_amd import synthetic
from deep3dmap_amd.multiview import MultiViewFit
g = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
v, tri = synthetic.grid_mesh(30)
tex = synthetic.random_textures(tri.shape[0], 2)
fit = MultiViewFit(v, tri, tex, synthetic.camera_ring(n), image_size=96, view_groups=g)
fit.set_targets_from(synthetic.perturb(v, 0.03))
if os.environ.get("D3M_DBG_MASKSUM"):
    fit.mask_sum = fit.targets[2].sum().reshape(1)
loss, gv, gt = fit.step()
print("eager", g, float(loss), flush=True)
fit.capture_graph()
print("captured", flush=True)
for _ in range(3):
    loss, gv, gt = fit.step()
torch.cuda.synchronize()
print("replayed", float(loss), flush=True)
