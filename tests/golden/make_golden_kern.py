#!/usr/bin/env python3
"""
Regenerates the KERNEL-LEVEL golden vectors from the reference's own kernels run on the device.  GPU BOX ONLY:
needs a GPU and oracle/_ref/libnr_ref_hip.so (the text of NR/cuda/rasterize_cuda_kernel.cu:22-593,
load_textures_cuda_kernel.cu:6-115 and create_texture_image_cuda_kernel.cu:8-116 compiled by hipcc where it lies,
-ffp-contract=off; `make -C oracle ref_hip` in the build container -- the .so travels, the reference does not).

  tests/golden/kern_golden.npz   kern/<case>/{inputs, every map the six rasterizer kernels write, gradients}
  tests/golden/tex_golden.npz    load/<case>/..., atlas/<case>/...   (uv coordinates avoid exact integers: there the
                                 reference's in-place wrap is a data race; `valid` = atlas pixels of real
                                 faces: the reference reads out of bounds for the padding tiles)

Stored: inputs and the reference kernels' numeric outputs only.  K5/K6 sum with float atomics, so grad_textures
and the depth part of grad_faces are reproducible to rounding, not to the bit; everything else is deterministic.

Usage (from the repository root, on the GPU box):
    python tests/golden/make_golden_kern.py --out gpurun_out/golden
and copy the two files into tests/golden/.
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import nr_oracle as O  # noqa: E402


def random_faces(rng, B, Fn, spread=1.2, zlo=0.5, zhi=3.0, size=0.6):
    """Random triangles in NDC: a centre in [-spread, spread]^2 plus offsets of scale `size`."""
    c = rng.uniform(-spread, spread, (B, Fn, 1, 2))
    xy = c + rng.uniform(-size, size, (B, Fn, 3, 2))
    z = rng.uniform(zlo, zhi, (B, Fn, 3, 1))
    return np.concatenate([xy, z], -1).astype(np.float32)


def kernel_case(rng, B, Fn, S, ts, near, far, eps, faces=None, bg=(0.1, 0.2, 0.3)):
    """Runs the six reference kernels (the reference's kernels on the device, back end "hip") in full mode and records every map."""
    faces = random_faces(rng, B, Fn) if faces is None else faces
    B, Fn = faces.shape[:2]
    tex = rng.uniform(0, 1, (B, Fn, ts, ts, ts, 3)).astype(np.float32)
    m = O.raster_forward(faces, tex, S, near, far, eps, bg, True, True, True, backend="hip")
    g_rgb = rng.normal(0, 1, (B, S, S, 3)).astype(np.float32)
    g_alpha = rng.normal(0, 1, (B, S, S)).astype(np.float32)
    g_depth = rng.normal(0, 1, (B, S, S)).astype(np.float32)
    out = dict(faces=faces, textures=tex, image_size=S, near=near, far=far, eps=eps, background=np.array(bg, np.float32),
               faces_inv=m["faces_inv"], face_index_map=m["face_index_map"], weight_map=m["weight_map"],
               depth_map=m["depth_map"], face_inv_map=m["face_inv_map"], rgb_map=m["rgb_map"],
               alpha_map=m["alpha_map"], sampling_index_map=m["sampling_index_map"],
               sampling_weight_map=m["sampling_weight_map"], grad_rgb_map=g_rgb, grad_alpha_map=g_alpha,
               grad_depth_map=g_depth)
    # the three backward kernels one by one (K4 alone, K5, K6 alone) and the composed order
    gf_all, gt = O.raster_backward(m, g_rgb, g_alpha, g_depth, True, True, True, backend="hip")
    gf_pix_rgba, _ = O.raster_backward(m, g_rgb, g_alpha, None, True, True, False, backend="hip")
    m_a = dict(m)
    gf_pix_alpha, _ = O.raster_backward(m_a, None, g_alpha, None, False, True, False, backend="hip")
    gf_depth, _ = O.raster_backward(m, None, None, g_depth, False, False, True, backend="hip")
    out.update(grad_faces_all=gf_all, grad_textures=gt, grad_faces_pixel_rgba=gf_pix_rgba,
               grad_faces_pixel_alpha=gf_pix_alpha, grad_faces_depth=gf_depth)
    return out


def rasterizer_cases():
    rng = np.random.default_rng(20261002)
    out = {}
    cases = {}
    cases["rand_b2_f24_s32_ts2"] = kernel_case(rng, 2, 24, 32, 2, 0.1, 100.0, 1e-3)
    cases["rand_b1_f64_s64_ts4"] = kernel_case(rng, 1, 64, 64, 4, 0.1, 100.0, 1e-4)
    # ts=1 (NrRenderer's tx_size=1): KCU:222-233 then indexes texels 1..3 of a 1-texel cube, i.e. the
    # following faces' texels (weights ~ -eps).  The last 3 faces are moved off screen so that this
    # bleed never leaves the buffer; in-buffer bleed is part of the recorded behaviour.
    f1 = random_faces(rng, 3, 12)
    f1[:, -3:, :, 0] += 10.0
    cases["rand_b3_f12_s16_ts1"] = kernel_case(rng, 3, 12, 16, 1, 0.1, 100.0, 1e-3, faces=f1)
    # near / far rejects: depth range wider than [near, far]
    cases["nearfar_b1_f32_s32"] = kernel_case(rng, 1, 32, 32, 2, 1.0, 2.0, 1e-3,
                                              faces=random_faces(rng, 1, 32, zlo=0.5, zhi=3.0))
    # ties: every face duplicated (same geometry at index f and f+16) -> lowest index must win;
    # plus a zero batch entry (all-zero vertices, as the reference's to_minibatch fixture produces)
    f = random_faces(rng, 1, 16)
    tie = np.concatenate([f, f], 1)
    tie = np.concatenate([tie, np.zeros_like(tie)], 0)
    cases["ties_zero_batch_b2_f32_s32"] = kernel_case(rng, 2, 32, 32, 2, 0.1, 100.0, 1e-3, faces=tie)
    # fill_back pairs (front + reversed winding), small triangles on a 48x48 raster
    f = random_faces(rng, 2, 40, size=0.15)
    cases["fillback_small_b2_f80_s48"] = kernel_case(rng, 2, 80, 48, 2, 0.1, 100.0, 1e-3,
                                                     faces=np.concatenate([f, f[:, :, ::-1, :]], 1).copy())
    # grid mesh whose vertices sit exactly on pixel centres / edges (shared-edge ties)
    n = 9
    gx, gy = np.meshgrid(np.linspace(-1, 1, n), np.linspace(-1, 1, n))
    verts = np.stack([gx, gy, 1.0 + 0.25 * np.sin(3 * gx) * np.cos(2 * gy)], -1).reshape(-1, 3).astype(np.float32)
    idx = np.arange(n * n).reshape(n, n)
    tri = np.concatenate([np.stack([idx[:-1, :-1], idx[:-1, 1:], idx[1:, :-1]], -1).reshape(-1, 3),
                          np.stack([idx[:-1, 1:], idx[1:, 1:], idx[1:, :-1]], -1).reshape(-1, 3)], 0)
    gf = verts[tri][None]
    cases["grid_on_pixel_centres_b1_f128_s32"] = kernel_case(rng, 1, gf.shape[1], 32, 2, 0.1, 100.0, 1e-3,
                                                             faces=np.concatenate([gf, gf[:, :, ::-1, :]], 1).copy())
    for cname, c in cases.items():
        for k, v in c.items():
            out[f"kern/{cname}/{k}"] = np.asarray(v)

    return out


def texture_cases():
    rng = np.random.default_rng(20261003)
    out = {}
    for wrapping in range(4):
        for bilinear in (1, 0):
            F, ts, H, W = 11, (4, 2, 3, 5)[wrapping], 13 + wrapping, 17 - wrapping
            image = rng.random((H, W, 3)).astype(np.float32)
            uv = rng.uniform(-1.75, 2.75, (F, 3, 2)).astype(np.float32)
            uv[0] = rng.uniform(0.05, 0.95, (3, 2))                 # an ordinary in-range face
            upd = (rng.random(F) > 0.3).astype(np.int32)
            upd[0] = 1
            tin = rng.random((F, ts, ts, ts, 3)).astype(np.float32)
            tout = O.load_textures_np(image, uv, tin.copy(), upd, wrapping, bilinear, backend="hip")
            c = f"load/w{wrapping}_b{bilinear}"
            out.update({f"{c}/image": image, f"{c}/faces_uv": uv, f"{c}/is_update": upd, f"{c}/textures_in": tin,
                        f"{c}/wrapping": np.int32(wrapping), f"{c}/bilinear": np.int32(bilinear),
                        f"{c}/textures_out": tout})
    for F, tsi, tso in ((7, 4, 16), (10, 2, 8), (1, 3, 5), (16, 4, 16), (17, 6, 4)):
        tex = rng.random((F, tsi, tsi, tsi, 3)).astype(np.float32)
        image, vertices = O.create_texture_image_np(tex, tso, backend="hip")
        c = f"atlas/f{F}_ts{tsi}_o{tso}"
        out.update({f"{c}/textures": tex, f"{c}/tso": np.int32(tso), f"{c}/image": image, f"{c}/vertices": vertices,
                    f"{c}/valid": O.texture_atlas_valid_mask(F, tso, image.shape)})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=HERE)
    args = ap.parse_args()
    assert O.have_ref_hip(), "oracle/_ref/libnr_ref_hip.so missing: run `make -C oracle ref_hip` in the build container"
    os.makedirs(args.out, exist_ok=True)
    for name, fn in (("kern_golden.npz", rasterizer_cases), ("tex_golden.npz", texture_cases)):
        out = fn()
        path = os.path.join(args.out, name)
        np.savez_compressed(path, **out)
        print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
