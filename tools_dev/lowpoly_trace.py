"""Per-launch times of one eager step of a coarse mesh (D3M_TRACE_LAUNCHES=1: every launch announced, synchronised and
timed on stderr).  The library latches the switch at its FIRST launch, so it is set before anything is launched; the
warm-up steps are therefore traced too -- read the last step's lines.  python tools_dev/lowpoly_trace.py <grid n>"""
import os
import sys

os.environ["D3M_TRACE_LAUNCHES"] = "1"          # before the first launch of the process (d3m_launch.h)
import torch  # noqa: E402

sys.path.insert(0, os.getcwd())
from deep3dmap_amd import synthetic  # noqa: E402
from deep3dmap_amd.multiview import MultiViewFit  # noqa: E402

n = int(sys.argv[1])
v, tri = synthetic.grid_mesh(n)
tex = synthetic.random_textures(tri.shape[0], 2)
fit = MultiViewFit(v, tri, tex, synthetic.camera_ring(1), image_size=256, anti_aliasing=True)
fit.set_targets_from(synthetic.perturb(v, 0.02))
for _ in range(3):
    fit.step()
torch.cuda.synchronize()
print("---- traced step ----", file=sys.stderr, flush=True)
fit.step()
torch.cuda.synchronize()
