"""Deterministic synthetic workloads for tests and bench.py.  numpy only.

grid_mesh(n): n x n vertices with exactly the topology of deep3dmap's get_face_idx
(deep3dmap/core/renderer/utils.py:74-78), 2(n-1)^2 triangles, laid on a "bumpy sphere":
n=64 -> 7,938 (gan2shape grid), n=164 -> 53,138 (~BFM), n=225 -> 100,352 (headline), n=709 -> 1,002,528.
"""
import math

import numpy as np


def grid_topology(n):
    """[2(n-1)^2, 3] int32: (tl, bl, tr) block then (tr, bl, br) block, row-major cells."""
    idx = np.arange(n * n, dtype=np.int32).reshape(n, n)
    f1 = np.stack([idx[:-1, :-1], idx[1:, :-1], idx[:-1, 1:]], -1).reshape(-1, 3)
    f2 = np.stack([idx[:-1, 1:], idx[1:, :-1], idx[1:, 1:]], -1).reshape(-1, 3)
    return np.concatenate([f1, f2], 0)


def _lowpass3(a):
    p = np.pad(a, 1, mode="edge")
    return sum(p[i:i + a.shape[0], j:j + a.shape[1]] for i in range(3) for j in range(3)) / 9.0


def grid_mesh(n, seed=0):
    """(vertices [n*n,3] f32, triangles [2(n-1)^2,3] i32): lat-long sphere, radius 1 + 0.1*noise (seed,
    3x3 low-pass), normalised the way neural_renderer's load_obj does (load_obj.py:155-159)."""
    rng = np.random.default_rng(seed)
    noise = _lowpass3(rng.standard_normal((n, n)))
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    lat = (i / (n - 1) - 0.5) * math.pi * 0.98          # stay clear of the poles (no zero-area fans)
    lon = j / (n - 1) * 2 * math.pi * (1 - 1.0 / n)     # open seam: no duplicated column
    r = 1.0 + 0.1 * noise
    v = np.stack([r * np.cos(lat) * np.sin(lon), r * np.sin(lat), r * np.cos(lat) * np.cos(lon)], -1).reshape(-1, 3)
    v = v - v.min(0)[None]
    v = v / np.abs(v).max()
    v = v * 2
    v = v - v.max(0)[None] / 2
    return v.astype(np.float32), grid_topology(n)


def icosphere(level=1):
    """(vertices [V,3], triangles [F,3]): level 1 = 42 vertices, 80 triangles (pt3d_demos config)."""
    t = (1.0 + 5 ** 0.5) / 2
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
         (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10),
         (8, 6, 7), (9, 8, 1)]
    v = [np.array(p, np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(level):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.array(v, np.float32), np.array(f, np.int32)


def camera_ring(n_views, distance=2.732):
    """eye of view i of n: get_points_from_angles(distance, 20 sin(2 pi i/n), 360 i/n) -> [n,3] f32."""
    eyes = []
    for i in range(n_views):
        el = math.radians(20.0 * math.sin(2 * math.pi * i / n_views))
        az = math.radians(360.0 * i / n_views)
        eyes.append((distance * math.cos(el) * math.sin(az), distance * math.sin(el),
                     -distance * math.cos(el) * math.cos(az)))
    return np.array(eyes, np.float32)


def random_textures(n_faces, texture_size=2, seed=1):
    return np.random.default_rng(seed).random((n_faces, texture_size, texture_size, texture_size, 3), dtype=np.float32)


def perturb(vertices, sigma=0.02, seed=2):
    return (vertices + np.random.default_rng(seed).normal(0, sigma, vertices.shape)).astype(np.float32)


def gan2shape_inputs(b, hw=64, seed=0, flip=False):
    """The tensors the gan2shape step hands to its renderer block (models/frameworks/gan2shape.py:418-460; SURVEY.md 8d,
    config C3): depth [B,hw,hw] = 1 + 0.1 tanh(low-passed noise) with the reference's border clamp (:428-430), albedo
    [B,3,hw,hw] and light [B,4] in (-1, 1), view vectors [B,6] ~ U(-1,1) x (0.5, 1.0, 0.3 rad; 0.1, 0.1, 0.02): the range :440-443 gives a
    tanh-bounded view head with configs/gan2shape/celeba.py's 60 degrees / 0.1, input image [b,3,hw,hw]; with `flip` the mirrored copies are appended (:431,449)."""
    rng = np.random.default_rng(seed)
    noise = np.pad(rng.standard_normal((b, hw, hw)), ((0, 0), (2, 2), (2, 2)))
    depth = 1.0 + 0.1 * np.tanh(sum(noise[:, i:i + hw, j:j + hw] for i in range(5) for j in range(5)) / 25.0)
    border = np.zeros((b, hw, hw))
    border[:, :, :2] = 1.02
    border[:, :, -2:] = 1.02
    depth = depth * (1 - border) + border * 1.04              # border_depth = 0.7 max + 0.3 min (gan2shape.py:58)
    albedo = np.tanh(rng.standard_normal((b, 3, hw, hw)))
    light = np.tanh(0.5 * rng.standard_normal((b, 4)))
    view = rng.uniform(-1, 1, (b, 6)) * np.array([0.5, 1.0, 0.3, 0.1, 0.1, 0.02])
    input_im = np.tanh(rng.standard_normal((b, 3, hw, hw)))
    if flip:
        depth = np.concatenate([depth, depth[:, :, ::-1]], 0)
        albedo = np.concatenate([albedo, albedo[:, :, :, ::-1]], 0)
        light, view = np.tile(light, (2, 1)), np.tile(view, (2, 1))
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return f(depth), f(albedo), f(light), f(view), f(input_im)
