import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_ops as T
from deep3dmap_amd.neural_renderer import rasterize_ops as ops
from oracle import nr_oracle as O
seed = int(sys.argv[1])
rng = np.random.default_rng(1000 + seed)
B = int(rng.integers(1, 3)); S = int(rng.choice([16, 24, 33, 48, 64])); Fn = int(rng.integers(1, 40))
size = float(rng.choice([0.03, 0.15, 0.6, 2.0]))
xy = rng.uniform(-1.2, 1.2, (B, Fn, 1, 2)) + rng.uniform(-size, size, (B, Fn, 3, 2))
mode = seed % 4
if mode == 1: xy = (2 * np.round((xy * S + S - 1) / 2) + 1 - S) / S
elif mode == 2: xy = np.round(xy * S / 2) * 2 / S
faces = np.concatenate([xy, rng.uniform(0.3, 4.0, (B, Fn, 3, 1))], -1).astype(np.float32)
if Fn > 3:
    faces[:, -1] = faces[:, 0]; faces[:, -2, 1] = faces[:, -2, 0]
faces = np.concatenate([faces, faces[:, :, ::-1]], 1).copy()
F2 = faces.shape[1]
tex = rng.uniform(0, 1, (B, F2, 2, 2, 2, 3)).astype(np.float32)
m = O.raster_forward(faces, tex, S, 0.5, 3.5, 1e-3, (0.1, 0.2, 0.3), True, True, True)
g_rgb = rng.normal(size=(B, S, S, 3)).astype(np.float32); g_alpha = rng.normal(size=(B, S, S)).astype(np.float32)
gf_ref, _ = O.raster_backward(m, g_rgb, g_alpha, None, True, True, False)
fd = T._dev(faces); gf = torch.zeros_like(fd)
ops.backward_pixel_map(fd, T._dev(m["face_index_map"]), T._dev(m["rgb_map"]), T._dev(m["alpha_map"]), T._dev(g_rgb), T._dev(g_alpha), gf, S, 1e-3, True, True)
g = gf.cpu().numpy()
print("B S Fn size mode", B, S, Fn, size, mode)
bad_ref = np.argwhere(~np.isfinite(gf_ref).all((2, 3))); bad = np.argwhere(~np.isfinite(g).all((2, 3)))
print("nonfinite faces ref:", bad_ref.tolist(), " hip:", bad.tolist())
for b, f in (bad_ref.tolist() + bad.tolist())[:3]:
    px = (faces[b, f, :, :2] * S + S - 1) / 2
    print("face", b, f, "pixel coords", px.tolist(), "ref", gf_ref[b, f].tolist(), "hip", g[b, f].tolist())
