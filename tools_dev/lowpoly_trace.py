import os, sys, torch
sys.path.insert(0, os.getcwd())
from deep3dmap_amd import synthetic
from deep3dmap_amd.multiview import MultiViewFit
n = int(sys.argv[1])
v, tri = synthetic.grid_mesh(n)
tex = synthetic.random_textures(tri.shape[0], 2)
fit = MultiViewFit(v, tri, tex, synthetic.camera_ring(1), image_size=256, anti_aliasing=True)
fit.set_targets_from(synthetic.perturb(v, 0.02))
for _ in range(3): fit.step()
torch.cuda.synchronize()
os.environ["D3M_TRACE_LAUNCHES"] = "1"
fit.step(); torch.cuda.synchronize()
