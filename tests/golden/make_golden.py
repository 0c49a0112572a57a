#!/usr/bin/env python3
"""
Regenerates tests/golden/nr_golden.npz.  BUILD CONTAINER ONLY: needs /root/reference.
(The kernel-level vectors -- kern_golden.npz, tex_golden.npz -- come from the reference's kernels run ON THE
DEVICE: tests/golden/make_golden_kern.py, GPU box.)

Sources of truth used here (nothing from them is stored except inputs and numeric outputs):
  * the reference's pure-torch modules, imported by file path from /root/reference
    (NR/{projection,perspective,look_at,vertices_to_faces,lighting,get_points_from_angles}.py,
    deep3dmap/core/renderer/utils.py, deep3dmap/core/utils/utils.py);
  * the known-answer constants of the reference's own tests
    (pnpmodules/neural_renderer/tests/test_rasterize_silhouettes.py:40-51,72-83, test_look_at.py:10-19,
    test_perspective.py:11-12).

Usage:  python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import nr_oracle as O  # noqa: E402

REF = "/root/reference"
NR = os.path.join(REF, "pnpmodules/neural_renderer/neural_renderer")


def load_ref(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    out = {}

    # ---- 2. known answers from the reference's own tests -------------------------------------
    out["known/sil1/vertices"] = np.array([[0.8, 0.8, 1.], [0.0, -0.5, 1.], [0.2, -0.4, 1.]], np.float32)
    out["known/sil1/pxy"] = np.array([35, 25])
    out["known/sil1/grad"] = np.array([[1.6725862, -0.26021874, 0.], [1.41986704, -1.64284933, 0.], [0., 0., 0.]], np.float32)
    out["known/sil2/vertices"] = np.array([[0.8, 0.8, 1.], [-0.5, -0.8, 1.], [0.8, -0.8, 1.]], np.float32)
    out["known/sil2/pxy"] = np.array([50, 40])
    out["known/sil2/grad"] = np.array([[0.98646867, 1.04628897, 0.], [-1.03415668, -0.10403691, 0.],
                                       [3.00094461, -1.55173182, 0.]], np.float32)
    out["known/look_at/eyes"] = np.array([[1, 0, 1], [0, 0, -10], [-1, 1, 0]], np.float32)
    out["known/look_at/answers"] = np.array([[-np.sqrt(2) / 2, 0, np.sqrt(2) / 2], [1, 0, 10],
                                             [0, np.sqrt(2) / 2, 3. / 2. * np.sqrt(2)]], np.float32)
    out["known/perspective/in"] = np.array([1, 2, 10], np.float32)
    out["known/perspective/out"] = np.array([np.sqrt(3) / 10, 2 * np.sqrt(3) / 10, 10], np.float32)

    # ---- 3. reference python modules (camera math, gather, lighting) --------------------------
    torch.manual_seed(7)
    r_proj = load_ref("ref_projection", os.path.join(NR, "projection.py")).projection
    r_persp = load_ref("ref_perspective", os.path.join(NR, "perspective.py")).perspective
    r_lookat = load_ref("ref_look_at", os.path.join(NR, "look_at.py")).look_at
    r_v2f = load_ref("ref_v2f", os.path.join(NR, "vertices_to_faces.py")).vertices_to_faces
    r_light = load_ref("ref_lighting", os.path.join(NR, "lighting.py")).lighting
    r_gpa = load_ref("ref_gpa", os.path.join(NR, "get_points_from_angles.py")).get_points_from_angles

    B, V = 3, 17
    verts = torch.randn(B, V, 3) * 0.5 + torch.tensor([0., 0., 2.5])
    K = torch.tensor([[[300., 0., 128.], [0., 310., 120.], [0., 0., 1.]]]).repeat(B, 1, 1)
    ang = torch.tensor([0.1, -0.2, 0.3])
    R = O.get_rotation_matrix(ang, ang * 0.5, -ang)
    t = torch.randn(B, 1, 3) * 0.1
    dist = torch.tensor([[0.1, -0.05, 0.01, 0.02, 0.003]]).repeat(B, 1)
    vg = verts.clone().requires_grad_(True)
    pr = r_proj(vg, K, R, t, dist, 256)
    gpr = torch.randn_like(pr)
    pr.backward(gpr)
    out.update({"cam/projection/vertices": verts.numpy(), "cam/projection/K": K.numpy(), "cam/projection/R": R.numpy(),
                "cam/projection/t": t.numpy(), "cam/projection/dist": dist.numpy(), "cam/projection/orig_size": np.array(256),
                "cam/projection/out": pr.detach().numpy(), "cam/projection/grad_out": gpr.numpy(),
                "cam/projection/grad_vertices": vg.grad.numpy()})
    # NB batch 4, not 3: NR/look_at.py:49-50 calls torch.cross without dim, which (legacy default) crosses
    # along the FIRST axis of size 3 -- the batch axis when batch_size == 3.  That accident is not part of
    # the contract ("Deviations"); the fixture stays clear of it.
    verts_la = torch.cat([verts, verts[:1] * 0.9 + 0.1], 0)
    eyes = torch.tensor([[0.5, 1.0, -2.5], [2.0, 0.3, 1.0], [-1.0, -0.7, -2.0], [0.0, 2.0, -1.5]])
    vg = verts_la.clone().requires_grad_(True)
    la = r_lookat(vg, eyes)
    pp = r_persp(la, angle=25.0)
    gpp = torch.randn_like(pp)
    pp.backward(gpp)
    out.update({"cam/look_at/vertices": verts_la.numpy(), "cam/look_at/eyes": eyes.numpy(),
                "cam/look_at/out": la.detach().numpy(), "cam/look_at/persp_angle": np.array(25.0),
                "cam/look_at/persp_out": pp.detach().numpy(), "cam/look_at/grad_out": gpp.numpy(),
                "cam/look_at/grad_vertices": vg.grad.numpy()})
    la1 = r_lookat(verts_la, [0, 0, -2.732])
    out["cam/look_at/single_eye_out"] = la1.numpy()
    out["cam/points_from_angles/in"] = np.array([[2.732, 20., 45.], [3.0, -10., 200.], [1.5, 0., 0.]], np.float32)
    out["cam/points_from_angles/out"] = np.array([r_gpa(float(d), float(e), float(a)) for d, e, a in
                                                  out["cam/points_from_angles/in"]], np.float32)
    fidx = torch.randint(0, V, (B, 11, 3), dtype=torch.int32)
    out["cam/v2f/faces"] = fidx.numpy()
    out["cam/v2f/out"] = r_v2f(verts, fidx).numpy()
    fl = r_v2f(verts, fidx)
    tex = torch.rand(B, 11, 2, 2, 2, 3)
    lit = r_light(fl, tex.clone(), 0.3, 0.7, (1.0, 0.9, 0.8), (0.5, 0.6, 1.0), (0.2, 0.9, -0.4))
    out.update({"cam/lighting/textures": tex.numpy(), "cam/lighting/out": lit.numpy(),
                "cam/lighting/params": np.array([0.3, 0.7, 1.0, 0.9, 0.8, 0.5, 0.6, 1.0, 0.2, 0.9, -0.4], np.float32)})

    # ---- 4. deep3dmap adapter helpers and losses ----------------------------------------------
    cu = load_ref("ref_cr_utils", os.path.join(REF, "deep3dmap/core/renderer/utils.py"))
    # deep3dmap/core/utils/utils.py pulls heavy optional imports at module top; give it only what it names
    for modname in ("cv2",):
        if modname not in sys.modules:
            try:
                __import__(modname)
            except Exception:
                sys.modules[modname] = types.ModuleType(modname)
    try:
        uu = load_ref("ref_core_utils", os.path.join(REF, "deep3dmap/core/utils/utils.py"))
    except Exception as e:  # pragma: no cover
        raise SystemExit(f"cannot import reference losses: {e}")
    view = torch.tensor([[0.1, -0.3, 0.05, 0.02, -0.01, 0.03], [-0.2, 0.4, 0.0, 0.0, 0.05, -0.02]])
    rot, tr = cu.get_transform_matrices(view)
    out.update({"d3m/view6": view.numpy(), "d3m/rot": rot.numpy(), "d3m/trans": tr.numpy()})
    rot5, tr5 = cu.get_transform_matrices(view[:, :5])
    out.update({"d3m/rot5": rot5.numpy(), "d3m/trans5": tr5.numpy()})
    out["d3m/grid_2_3_4_raw"] = cu.get_grid(2, 3, 4, normalize=False).numpy()
    out["d3m/grid_1_3_4_norm"] = cu.get_grid(1, 3, 4, normalize=True).numpy()
    out["d3m/face_idx_2_3_4"] = cu.get_face_idx(2, 3, 4).numpy()
    im = torch.rand(2, 3, 4, 5)
    out["d3m/im"] = im.numpy()
    out["d3m/tex_from_im_ts1"] = cu.get_textures_from_im(im, tx_size=1).numpy()
    out["d3m/tex_from_im_ts2"] = cu.get_textures_from_im(im, tx_size=2).numpy()
    a, b = torch.rand(2, 3, 8, 8), torch.rand(2, 3, 8, 8)
    mask = (torch.rand(2, 1, 8, 8) > 0.3).float()
    sig = torch.rand(2, 1, 8, 8) + 0.1
    out.update({"loss/a": a.numpy(), "loss/b": b.numpy(), "loss/mask": mask.numpy(), "loss/sigma": sig.numpy(),
                "loss/photometric": uu.photometric_loss(a, b).numpy(),
                "loss/photometric_mask": uu.photometric_loss(a, b, mask=mask).numpy(),
                "loss/photometric_mask_sigma": uu.photometric_loss(a, b, mask=mask, conf_sigma=sig).numpy(),
                "loss/smooth": uu.smooth_loss(a[:, 0]).numpy(),
                "loss/smooth_pyramid": uu.smooth_loss([a[:, 0], b[:, 0, ::2, ::2]]).numpy()})

    path = os.path.join(HERE, "nr_golden.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
