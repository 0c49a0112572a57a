#!/usr/bin/env python3
"""
Regenerates tests/golden/mesh_golden.npz.  BUILD CONTAINER ONLY: needs oracle/_ref/libmesh_ref.so, i.e. the
reference's own face3d rasterizer (deep3dmap/core/renderer/renderer_demo/mesh_cython/render.cpp) compiled from its
sources by `make -C oracle ref`.  Stored: the outputs of every function of the family on the deterministic scenes of
tests/mesh_scenes.py (inputs are regenerated from the seeds, not stored).

Usage:  python tests/golden/make_golden_mesh.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mesh_scenes  # noqa: E402
from oracle import mesh_oracle as M  # noqa: E402


class _Ref:
    def __getattr__(self, name):
        fn = getattr(M, name)
        return lambda *a, **k: fn(*a, backend="ref", **k)


def main():
    assert M.have_ref(), "run `make -C oracle ref` first"
    out = {}
    for name, (n, h, w, seed) in mesh_scenes.SCENES.items():
        res = mesh_scenes.run_family(_Ref(), mesh_scenes.grid_scene(n, h, w, seed))
        for k, v in res.items():
            out[f"{name}/{k}"] = np.asarray(v)
    path = os.path.join(HERE, "mesh_golden.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
