"""which torch op launches the one copy kernel per gan2shape step?"""
import torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from deep3dmap_amd import synthetic
from deep3dmap_amd.core import NrRenderer
from torch.profiler import profile, ProfilerActivity
b, hw = 16, 64
r = NrRenderer(dict(min_depth=0.9, max_depth=1.1, rot_center_depth=1.0, fov=10, tex_cube_size=2), hw)
depth, albedo, light, view, im = (torch.from_numpy(a).cuda() for a in synthetic.gan2shape_inputs(b, hw, 0, False))
la, lb = light[:, :1] / 2 + 0.5, light[:, 1:2] / 2 + 0.5
ld = torch.cat([light[:, 2:], torch.ones(b, 1, device="cuda")], 1); ld = ld / ((ld ** 2).sum(1, keepdim=True)) ** 0.5
leaves = [t.detach().clone().requires_grad_(True) for t in (depth, albedo, la, lb, ld, view)]
d, a, la, lb, ld, v = leaves
one = torch.ones((), device="cuda")
def step():
    for x in leaves: x.grad = None
    r.set_transform_matrices(v)
    out = r.reconstruct(d, a, la, lb, ld, im, flip=False, lam_smooth=0.01)
    torch.autograd.backward([out.loss_total], [one])
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy") :
        print(e.name, e.input_shapes, [s for s in (e.stack or [])][:6])
