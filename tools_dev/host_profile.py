"""on the GPU box: where the HOST time of an eager (not captured) fit step goes -- cProfile over 300 steps of BASELINE config 2"""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.getcwd())
from deep3dmap_amd import synthetic
from deep3dmap_amd.multiview import MultiViewFit
v, tri = synthetic.grid_mesh(164)
tex = synthetic.random_textures(tri.shape[0], 2)
fit = MultiViewFit(v, tri, tex, synthetic.camera_ring(1), image_size=256, anti_aliasing=True)
fit.set_targets_from(synthetic.perturb(v, 0.02))
for _ in range(20):
    fit.step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(300):
    fit.step()
torch.cuda.synchronize()
print("eager ms/step", (time.perf_counter() - t0) / 300 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    fit.step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
