"""How often does the known deviation of the binned forward (a near-zero-area sliver selected by brute force
at a pixel outside its dilated bounding box) occur?  Scenes of slivers, product vs the reference's kernels on the device."""
import numpy as np
import torch
from oracle import nr_ref_hip as RH
from tests.test_gpu_reference import _product_forward

tot_px = tot_bad = tot_faces = 0
for seed in range(60):
    rng = np.random.default_rng(seed)
    S = int(rng.choice([32, 64, 128]))
    Fn = 400
    p0 = rng.uniform(-1.1, 1.1, (1, Fn, 1, 2))
    d = rng.uniform(-1, 1, (1, Fn, 1, 2)) * float(rng.choice([0.05, 0.3, 1.5]))
    t = rng.uniform(0, 1, (1, Fn, 3, 1))
    xy = p0 + d * t                                          # three collinear points ...
    kind = seed % 3
    if kind == 0:
        xy = xy + rng.normal(size=xy.shape) * 1e-7           # ... up to rounding
    elif kind == 1:
        xy = xy + rng.normal(size=xy.shape) * 1e-4           # ... thin slivers
    else:                                                    # ... axis-aligned or diagonal through pixel centres, exact
        c = (2 * rng.integers(0, S, (1, Fn, 1, 2)) + 1 - S) / S
        step = rng.integers(-3, 4, (1, Fn, 1, 2)) * 2.0 / S
        xy = c + step * rng.integers(0, 6, (1, Fn, 3, 1))
    faces = np.concatenate([xy, rng.uniform(0.6, 3.0, (1, Fn, 3, 1))], -1).astype(np.float32)
    faces = np.concatenate([faces, faces[:, :, ::-1]], 1).copy()
    fd = torch.from_numpy(faces).cuda()
    td = torch.rand(1, faces.shape[1], 2, 2, 2, 3, device="cuda")
    ref = RH.forward(fd, td, S, 0.5, 3.5, 1e-3, (0, 0, 0))
    m = _product_forward(fd, td, S, 0.5, 3.5, 1e-3, (0, 0, 0))
    bad = int((m["face_index_map"] != ref["face_index_map"]).sum())
    cov = int((ref["face_index_map"] >= 0).sum())
    tot_px += S * S; tot_bad += bad; tot_faces += faces.shape[1]
    if bad:
        print("seed", seed, "kind", kind, "S", S, "mismatched", bad, "covered", cov)
print("scenes 60, faces", tot_faces, "pixels", tot_px, "mismatched", tot_bad)
