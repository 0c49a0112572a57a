import sys, os, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep3dmap_amd import synthetic
from deep3dmap_amd.multiview import MultiViewFit
v, tri = synthetic.grid_mesh(30)
tex = synthetic.random_textures(tri.shape[0], 2)
keep = []
for g in [int(a) for a in sys.argv[1].split(",")]:
    fit = MultiViewFit(v, tri, tex, synthetic.camera_ring(5), image_size=96, view_groups=g)
    if os.environ.get("NODEFER"):
        fit.renderer.defer_plan_join = False
    fit.set_targets_from(synthetic.perturb(v, 0.03))
    loss, gv, gt = fit.step()
    fit.capture_graph()
    for _ in range(3):
        loss, gv, gt = fit.step()
    torch.cuda.synchronize()
    print("ok", g, float(loss), flush=True)
    if os.environ.get("KEEP"):
        keep.append(fit)
