"""on the GPU box: run-to-run spread of the headline step's results.  The visibility list's chunks land in arrival order
(k_compact_visible_atomic), so the float atomics of the passes that run over it add in another order every run: the
gradients agree to rounding, not bit for bit.  Writes gpurun_out/final/<R>_run_to_run_spread.json."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from deep3dmap_amd import synthetic
from deep3dmap_amd.multiview import MultiViewFit
R = os.environ.get("R", "r05")
out = {}
for views, size in ((32, 512), (4, 512)):
    v, tri = synthetic.grid_mesh(225)
    tex = synthetic.random_textures(tri.shape[0], 2)
    fit = MultiViewFit(v, tri, tex, synthetic.camera_ring(views), image_size=size)
    fit.set_targets_from(synthetic.perturb(v))
    runs = []
    for mode in ("eager", "graph"):
        if mode == "graph":
            fit.capture_graph()
        for _ in range(6):
            loss, gv, gt = fit.step()
            runs.append((float(loss), gv.clone(), gt.clone()))
    fit.release_graph()
    l0, gv0, gt0 = runs[0]
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    out[f"{views}views_{size}"] = {
        "runs": len(runs), "loss_values_distinct": len({r[0] for r in runs}),
        "loss_max_rel_dev": max(abs(r[0] - l0) / abs(l0) for r in runs),
        "grad_vertices_max_dev_over_max_entry": max(rel(r[1], gv0) for r in runs),
        "grad_textures_max_dev_over_max_entry": max(rel(r[2], gt0) for r in runs),
        "grad_vertices_bit_identical_runs": sum(int(torch.equal(r[1], gv0)) for r in runs),
        "grad_textures_bit_identical_runs": sum(int(torch.equal(r[2], gt0)) for r in runs)}
    del fit
os.makedirs(os.path.join(ROOT, "gpurun_out", "final"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "final", f"{R}_run_to_run_spread.json"), "w"), indent=1)
print(json.dumps(out))
