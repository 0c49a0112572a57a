"""GPU tests of deep3dmap's adapter layer on the HIP path -- NrRenderer (deep3dmap/core/renderer/renderer_nr.py:12-277)
and its helpers (deep3dmap/core/renderer/utils.py:22-107) -- against the committed vectors made by the reference's own
utils.py (tests/golden/make_golden.py, d3m/*) and against the oracle's method-for-method restatement."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = dict(min_depth=0.9, max_depth=1.1, rot_center_depth=1.0, fov=10, tex_cube_size=2)


def _pair(hw, cfg=CFG):
    from deep3dmap_amd.core import NrRenderer
    from oracle import nr_oracle as O
    return NrRenderer(dict(cfg), hw), O.NrRenderer(dict(cfg), hw)


def _scene(b, hw, seed=0):
    rng = np.random.default_rng(seed)
    noise = torch.from_numpy(rng.standard_normal((b, hw, hw)).astype(np.float32))
    depth = 1.0 + 0.08 * torch.tanh(torch.nn.functional.avg_pool2d(noise[:, None], 5, 1, 2)[:, 0])
    im = torch.from_numpy(rng.uniform(-1, 1, (b, 3, hw, hw)).astype(np.float32))
    view = torch.from_numpy(rng.uniform(-1, 1, (b, 6)).astype(np.float32)) * torch.tensor([0.3, 0.5, 0.2, 0.05, 0.05, 0.02])
    return depth, im, view


def _close(a, b, tol=1e-5):
    a, b = a.detach().cpu(), b.detach().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = float((a - b).abs().max()) / max(1.0, float(b.abs().max()))
    assert err <= tol, err


def _images_close(a, b, frac=0.004, tol=2e-3):
    """rendered MESH frames: the product composes a frame's rigid motions into ONE (A, t) on [B,3,3] before it moves the
    points, the oracle moves them step by step as the reference does -- the vertices agree to an ulp or two, and an edge
    pixel whose centre lies that close to an edge may flip.  Everything else must agree to `tol`; the flipped fraction is
    bounded (measured: <= 0.2 % of the pixels at 32x32 .. 64x64)."""
    a, b = a.detach().cpu(), b.detach().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    bad = ((a - b).abs() > tol).float().mean()
    assert float(bad) <= frac, float(bad)


def _resampled_close(a, b, tol=5e-4):
    """grid_sample frames with the SAME (R, t) on both sides (_same_view_on_both_sides): identical vertices, identical
    coverage, so every pixel must agree -- to the f32 noise of a sampling coordinate scaled by the focal length"""
    a, b = a.detach().cpu(), b.detach().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert float((a - b).abs().max()) <= tol, float((a - b).abs().max())


def _same_view_on_both_sides(rg, ro):
    """set_transform_matrices on the HIP renderer from the ORACLE's (R, t) (host trigonometry): the two then move
    bit-identical vertices, rasterize the same coverage, and images can be held to float noise instead of a fraction of
    flipped edge pixels"""
    from oracle import nr_oracle as O

    def set_view(view):
        rot, trans = O.get_transform_matrices(view.detach().cpu())
        rg.rot_mat, rg.trans_xyz = rot.cuda(), trans.cuda()
    rg.set_transform_matrices = set_view


# ---- a18: helpers against the reference's own outputs (golden d3m/*) --------------------------------------------------
def test_helpers_match_reference_vectors(golden):
    from deep3dmap_amd.core import renderer_utils as U
    t = lambda k: torch.from_numpy(golden[k])
    assert torch.equal(U.get_grid(2, 3, 4, normalize=False, device="cuda").cpu(), t("d3m/grid_2_3_4_raw"))
    assert torch.allclose(U.get_grid(1, 3, 4, normalize=True, device="cuda").cpu(), t("d3m/grid_1_3_4_norm"))
    assert torch.equal(U.get_face_idx(2, 3, 4, device="cuda").cpu(), t("d3m/face_idx_2_3_4"))
    im = t("d3m/im").cuda()
    for ts in (1, 2):
        got = U.get_textures_from_im(im, tx_size=ts)
        ref = t(f"d3m/tex_from_im_ts{ts}")
        assert got.shape == ref.shape and torch.allclose(got.cpu(), ref, atol=1e-6)
    with pytest.raises(NotImplementedError):
        U.get_textures_from_im(im, tx_size=3)
    # the HIP Euler-angle kernel (d3m_view_transform) for 6- and 5-component views
    rot, trans = U.get_transform_matrices(t("d3m/view6").cuda())
    assert torch.allclose(rot.cpu(), t("d3m/rot"), atol=1e-6) and torch.allclose(trans.cpu(), t("d3m/trans"), atol=1e-7)
    rot5, trans5 = U.get_transform_matrices(t("d3m/view6")[:, :5].cuda())
    assert torch.allclose(rot5.cpu(), t("d3m/rot5"), atol=1e-6) and torch.allclose(trans5.cpu(), t("d3m/trans5"), atol=1e-7)
    # vcolor_to_texture_cube: the 8x3 coefficient matrix (utils.py:84-93) through the product's helper
    vc = torch.rand(2, 3, 5, 3)
    from oracle import nr_oracle as O
    assert torch.allclose(U.vcolor_to_texture_cube(vc.cuda()).cpu(), O.vcolor_to_texture_cube(vc), atol=1e-6)


@pytest.mark.parametrize("ts", [1, 2])
def test_textures_from_im_gradient(ts):
    from deep3dmap_amd.core import renderer_utils as U
    from oracle import nr_oracle as O
    im = torch.rand(2, 3, 6, 7)
    g = torch.randn(2, 2 * 5 * 6, ts, ts, ts, 3)
    a = im.clone().requires_grad_(True)
    O.get_textures_from_im(a, tx_size=ts).backward(g)
    b = im.clone().cuda().requires_grad_(True)
    U.get_textures_from_im(b, tx_size=ts).backward(g.cuda())
    _close(b.grad, a.grad, 1e-6)


# ---- a17: grids, warps, normals ------------------------------------------------------------------------------------
def test_grids_and_warps_against_oracle():
    hw, b = 24, 3
    rg, ro = _pair(hw)
    depth, _, view = _scene(b, hw, 1)
    rg.set_transform_matrices(view.cuda())
    ro.set_transform_matrices(view)
    dg = depth.cuda()
    for name in ("depth_to_3d_grid", "get_warped_3d_grid", "get_inv_warped_3d_grid", "get_warped_2d_grid",
                 "get_inv_warped_2d_grid", "get_normal_from_depth"):
        _close(getattr(rg, name)(dg), getattr(ro, name)(depth), 2e-5)
    g3 = ro.get_warped_3d_grid(depth)
    _close(rg.grid_3d_to_2d(g3.cuda()), ro.grid_3d_to_2d(g3), 2e-5)
    pts = g3.reshape(b, -1, 3)
    _close(rg.rotate_pts(pts.cuda(), rg.rot_mat), ro.rotate_pts(pts, ro.rot_mat), 1e-5)
    _close(rg.translate_pts(pts.cuda(), rg.trans_xyz), ro.translate_pts(pts, ro.trans_xyz), 1e-6)
    rg.downscale_K(2)
    ro.downscale_K(2)
    _close(rg.K, ro.K, 1e-6)
    _close(rg.inv_K, ro.inv_K, 1e-5)
    _close(rg.get_inv_warped_2d_grid(dg[:, ::2, ::2].contiguous()), ro.get_inv_warped_2d_grid(depth[:, ::2, ::2]), 2e-5)


@pytest.mark.parametrize("name", ["get_warped_3d_grid", "get_inv_warped_2d_grid", "get_normal_from_depth"])
def test_warp_gradients_against_oracle_autograd(name):
    """gradients wrt the depth map and (through the composed rigid transform) the view vector."""
    hw, b = 16, 2
    rg, ro = _pair(hw)
    depth, _, view = _scene(b, hw, 2)
    outs = []
    for r, dev in ((ro, "cpu"), (rg, "cuda")):
        d = depth.clone().to(dev).requires_grad_(True)
        v = view.clone().to(dev).requires_grad_(True)
        r.set_transform_matrices(v)
        out = getattr(r, name)(d)
        w = torch.linspace(-1, 1, out.numel()).reshape(out.shape).to(dev)
        (out * w).sum().backward()
        outs.append((d.grad.cpu(), v.grad.cpu() if v.grad is not None else None))
    (gd0, gv0), (gd1, gv1) = outs
    _close(gd1, gd0, 1e-4)
    if name != "get_normal_from_depth":
        _close(gv1, gv0, 1e-4)


# ---- a17: view synthesis -----------------------------------------------------------------------------------------
def test_render_given_view_both_branches():
    hw, b = 32, 2
    rg, ro = _pair(hw)
    depth, im, view = _scene(b, hw, 3)
    mask = (torch.rand(b, 1, hw, hw) > 0.3).float()
    _same_view_on_both_sides(rg, ro)          # (the mesh branch composes its own Rigid from the view: not affected)
    for grid_sample in (True, False):
        close = _resampled_close if grid_sample else _images_close
        a = ro.render_given_view(im, depth, view, grid_sample=grid_sample)
        g = rg.render_given_view(im.cuda(), depth.cuda(), view.cuda(), grid_sample=grid_sample)
        close(g, a)
        a_im, a_m = ro.render_given_view(im, depth, view, mask=mask if grid_sample else mask.repeat(1, 3, 1, 1),
                                         grid_sample=grid_sample)
        g_im, g_m = rg.render_given_view(im.cuda(), depth.cuda(), view.cuda(),
                                         mask=(mask if grid_sample else mask.repeat(1, 3, 1, 1)).cuda(), grid_sample=grid_sample)
        close(g_im, a_im)
        if grid_sample:     # 'nearest' flips where the coordinate noise crosses a texel boundary
            assert float((g_m.cpu() != a_m).float().mean()) < 2e-3
        else:
            _images_close(g_m, a_m)


def test_render_yaw_and_render_view():
    hw, b = 32, 2
    rg, ro = _pair(hw)
    depth, im, view = _scene(b, hw, 4)
    dg, ig = depth.cuda(), im.cuda()
    rot = torch.tensor([-0.4, 0.3])
    # mesh path: crop_mesh, inverse warp before, a view after (shared [b,6] and per-frame [t,b,6])
    cases = [dict(rotations=rot, crop_mesh=(2, 1, 3, 0)),
             dict(rotations=rot, v_before=view),
             dict(rotations=rot, v_after=view * 0.5),
             dict(rotations=rot, v_before=view, v_after=torch.stack([view * 0.3, view * -0.2]))]
    for kw in cases:
        a = ro.render_yaw(im, depth, **kw)
        g = rg.render_yaw(ig, dg, **{k: (v.cuda() if torch.is_tensor(v) and k != "rotations" else v) for k, v in kw.items()})
        assert g.shape == (b, 2, 3, hw, hw)
        _images_close(g, a)
    # grid_sample path, same (R, t) on both sides: every pixel
    _same_view_on_both_sides(rg, ro)
    a = ro.render_yaw(im, depth, rotations=rot, v_before=view, grid_sample=True)
    g = rg.render_yaw(ig, dg, rotations=rot, v_before=view.cuda(), grid_sample=True)
    _resampled_close(g, a)
    # default sweep length
    assert rg.render_yaw(ig, dg, nsample=3, maxr=30).shape == (b, 3, 3, hw, hw)
    for grid_sample in (False, True):
        a = ro.render_view(im, depth, v_before=view, maxr=[10, 30], nsample=[2, 3], grid_sample=grid_sample)
        g = rg.render_view(ig, dg, v_before=view.cuda(), maxr=[10, 30], nsample=[2, 3], grid_sample=grid_sample)
        assert g.shape == (b, 5, 3, hw, hw)
        (_resampled_close if grid_sample else _images_close)(g, a)


def test_crop_mesh_has_no_gradient_and_says_so():
    """render_yaw(crop_mesh=...) with grad-enabled inputs renders (as the reference does) and warns that the cropped mesh
    carries no gradient -- it does not raise"""
    rg, ro = _pair(16)
    depth, im, _ = _scene(1, 16, 5)
    with pytest.warns(RuntimeWarning, match="crop_mesh"):
        out = rg.render_yaw(im.cuda(), depth.cuda().requires_grad_(True), nsample=1, crop_mesh=(1, 1, 1, 1))
    assert out.shape == (1, 1, 3, 16, 16) and not out.requires_grad
    _images_close(out, ro.render_yaw(im, depth, nsample=1, crop_mesh=(1, 1, 1, 1)))


def test_gan2shape_step_batch16_against_oracle():
    """BASELINE config 3 at its batch size: depth [16,64,64] -> 7,938-triangle grid mesh (fill_back) -> render_depth
    @64 with anti-aliasing -> photometric + smooth loss -> gradients to the depth map and the view."""
    from deep3dmap_amd.core import photometric_loss, smooth_loss
    from oracle import nr_oracle as O
    hw, b = 64, 16
    rg, ro = _pair(hw)
    depth, _, view = _scene(b, hw, 6)
    target = torch.full((b, hw, hw), 1.0)
    # both sides start from the same (R, t): see test_gan2shape_step_warp_canon_depth_against_oracle
    rot0, trans0 = O.get_transform_matrices(view)
    outs = []
    for r, dev, P, S in ((ro, "cpu", O.photometric_loss, O.smooth_loss), (rg, "cuda", photometric_loss, smooth_loss)):
        d = depth.clone().to(dev).requires_grad_(True)
        r.rot_mat = rot0.clone().to(dev).requires_grad_(True)
        r.trans_xyz = trans0.clone().to(dev).requires_grad_(True)
        warped = r.warp_canon_depth(d)
        loss = P(warped[:, None], target.to(dev)[:, None]) + 0.01 * S(d)
        loss.backward()
        outs.append((warped.detach().cpu(), float(loss.detach()), d.grad.cpu(), r.rot_mat.grad.cpu(), r.trans_xyz.grad.cpu()))
    (w0, l0, g0, r0, t0), (w1, l1, g1, r1, t1) = outs
    rel_max = lambda a, b_: float((a - b_).abs().max() / b_.abs().max())
    assert rel_max(w1, w0) < 1e-5 and abs(l1 - l0) < 1e-5 * abs(l0)
    assert rel_max(g1, g0) < 1e-3 and rel_max(r1, r0) < 1e-3 and rel_max(t1, t0) < 1e-3


def test_grid_sample_frames_values_and_gradients_on_the_hip_resampler():
    """render_given_view(grid_sample=True) (CR:258-267): warp_canon_depth -> inverse-warped grid -> bilinear / nearest
    lookups, the last three as ONE HIP pass (d3m_warp_resample) instead of torch's grid_sample: values and the gradients
    wrt the image and the depth map against the oracle (torch-CPU grid_sample)."""
    hw, b = 32, 3
    rg, ro = _pair(hw)
    _same_view_on_both_sides(rg, ro)
    depth, im, view = _scene(b, hw, 8)
    mask = (torch.rand(b, 2, hw, hw, generator=torch.Generator().manual_seed(1)) > 0.3).float()
    w = torch.randn(b, 3, hw, hw, generator=torch.Generator().manual_seed(2))
    outs = []
    for r, dev in ((ro, "cpu"), (rg, "cuda")):
        d = depth.clone().to(dev).requires_grad_(True)
        i = im.clone().to(dev).requires_grad_(True)
        out, out_m = r.render_given_view(i, d, view.to(dev), mask=mask.to(dev), grid_sample=True)
        (out * w.to(dev)).sum().backward()
        outs.append((out.detach().cpu(), out_m.detach().cpu(), i.grad.cpu(), d.grad.cpu()))
    (o0, m0, gi0, gd0), (o1, m1, gi1, gd1) = outs
    # the sampling position is an f32 image coordinate with a focal length of ~5.7 image widths: ~1e-4 texel of noise
    assert float((o1 - o0).abs().max()) < 5e-4
    assert float((m1 != m0).float().mean()) < 2e-3          # 'nearest' flips where that noise crosses a texel boundary
    rel = lambda x, y: float((x - y).abs().max() / y.abs().max())
    assert rel(gi1, gi0) < 1e-3 and rel(gd1, gd0) < 2e-3
    # no torch grid_sample on the product's path
    import inspect
    from deep3dmap_amd.core import renderer_nr
    assert "functional.grid_sample" not in inspect.getsource(renderer_nr)


def test_image_grid_frames_equal_the_materialised_arrays_bit_for_bit():
    """render_rgb_image_grid (implicit topology, texture cubes evaluated from the image in the sampler) against
    render_rgb(vertices, get_face_idx, get_textures_from_im(im, 2)) -- CR:196-198 -- on the same vertices: same bits."""
    from deep3dmap_amd.core.renderer_nr import Rigid
    from deep3dmap_amd.core.renderer_utils import get_face_idx, get_textures_from_im
    for hw, b in ((24, 2), (33, 3)):
        rg, _ = _pair(hw)
        depth, im, view = _scene(b, hw, 12)
        with torch.no_grad():
            vertices = rg._warp(depth.cuda(), Rigid.of_view(view.cuda()))
            direct = rg.renderer.render_rgb_image_grid(vertices, im.cuda())
            ref = rg.renderer.render_rgb(vertices, get_face_idx(b, hw, hw, "cuda"), get_textures_from_im(im.cuda(), tx_size=2))
        assert direct.shape == ref.shape == (b, 3, hw, hw)
        assert torch.equal(direct, ref)
        assert float((direct - 1.0).abs().max()) > 0.1          # not an all-background image
    # with gradients wanted the frames take the materialised route and differentiate
    rg, _ = _pair(16)
    depth, im, view = _scene(1, 16, 13)
    i = im.cuda().requires_grad_(True)
    rg.render_given_view(i, depth.cuda(), view.cuda(), grid_sample=False).sum().backward()
    assert i.grad is not None and float(i.grad.abs().sum()) > 0
