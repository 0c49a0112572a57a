#!/usr/bin/env python3
"""
Regenerates tests/golden/tex_golden.npz.  BUILD CONTAINER ONLY: needs oracle/_ref/libnr_ref.so
(`make -C oracle ref`, which host-compiles the reference's load_textures / create_texture_image kernel
bodies where they lie: NR/cuda/load_textures_cuda_kernel.cu:6-115, create_texture_image_cuda_kernel.cu:8-116).

Stored: inputs and the reference kernels' numeric outputs only.
  load/<case>/{image,faces_uv,is_update,textures_in,wrapping,bilinear,textures_out}
  atlas/<case>/{textures,tso,image,vertices,valid}   (valid = atlas pixels that belong to a real face; the
                                                      reference reads out of bounds for the padding tiles)
uv coordinates avoid exact integers: there the reference's in-place wrap is a data race.

Usage:  python tests/golden/make_golden_textures.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import nr_oracle as O  # noqa: E402


def main():
    assert O.have_ref(), "run `make -C oracle ref` first"
    rng = np.random.default_rng(20261003)
    out = {}
    n = 0
    for wrapping in range(4):
        for bilinear in (1, 0):
            F, ts, H, W = 11, (4, 2, 3, 5)[wrapping], 13 + wrapping, 17 - wrapping
            image = rng.random((H, W, 3)).astype(np.float32)
            uv = rng.uniform(-1.75, 2.75, (F, 3, 2)).astype(np.float32)
            uv[0] = rng.uniform(0.05, 0.95, (3, 2))                 # an ordinary in-range face
            upd = (rng.random(F) > 0.3).astype(np.int32)
            upd[0] = 1
            tin = rng.random((F, ts, ts, ts, 3)).astype(np.float32)
            tout = O.load_textures_np(image, uv, tin.copy(), upd, wrapping, bilinear, backend="ref")
            c = f"load/w{wrapping}_b{bilinear}"
            out.update({f"{c}/image": image, f"{c}/faces_uv": uv, f"{c}/is_update": upd, f"{c}/textures_in": tin,
                        f"{c}/wrapping": np.int32(wrapping), f"{c}/bilinear": np.int32(bilinear),
                        f"{c}/textures_out": tout})
            n += 1
    for F, tsi, tso in ((7, 4, 16), (10, 2, 8), (1, 3, 5), (16, 4, 16), (17, 6, 4)):
        tex = rng.random((F, tsi, tsi, tsi, 3)).astype(np.float32)
        image, vertices = O.create_texture_image_np(tex, tso, backend="ref")
        c = f"atlas/f{F}_ts{tsi}_o{tso}"
        out.update({f"{c}/textures": tex, f"{c}/tso": np.int32(tso), f"{c}/image": image, f"{c}/vertices": vertices,
                    f"{c}/valid": O.texture_atlas_valid_mask(F, tso, image.shape)})
        n += 1
    path = os.path.join(HERE, "tex_golden.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {n} cases, {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
