"""GPU tests of the other BASELINE.json configurations and of the size-independent properties of the
full-size workload: gan2shape step (NrRenderer), pt3d_demos plumbing, multi-view fit, large faces."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rel_l2(a, b):
    return float(torch.linalg.norm(a - b) / (torch.linalg.norm(b) + 1e-12))


def test_gan2shape_step_warp_canon_depth_against_oracle():
    """BASELINE config 3 (reduced batch): 64x64 canonical depth -> grid mesh (7938 tris) -> render_depth with
    anti-aliasing through NrRenderer built from a gan2shape-style config, + photometric & smooth loss, backward."""
    from deep3dmap_amd.config import build_renderer, load_config
    from deep3dmap_amd.core import photometric_loss, smooth_loss
    from oracle import nr_oracle as O
    cfg = load_config(os.path.join(ROOT, "tests", "fixtures", "gan2shape_like.py"))
    cfg.model.model_cfgs["image_size"] = 64
    rg = build_renderer(cfg)
    ro = O.NrRenderer(cfg.model.model_cfgs, 64)
    rng = np.random.default_rng(0)
    b = 4
    noise = rng.standard_normal((b, 64, 64)).astype(np.float32)
    depth0 = torch.from_numpy(1.0 + 0.1 * np.tanh(noise * 0.3))
    # smooth the noise a little so that the mesh does not fold over itself
    depth0 = torch.nn.functional.avg_pool2d(depth0[:, None], 5, 1, 2)[:, 0]
    view = torch.from_numpy(rng.uniform(-1, 1, (b, 6)).astype(np.float32)) * torch.tensor([0.3, 0.5, 0.2, 0.05, 0.05, 0.02])
    target = torch.full((b, 64, 64), 1.0)
    # gan2shape learns the view (gan2shape.py:440-444).  Euler angles -> (R, t) goes through sin / cos, whose last bits
    # differ between libm and the device; that map has its own tests (d3m_view_transform vs the reference's vectors and
    # autograd), so here both sides start from the SAME (R, t) -- the oracle's grid warp computes in the HIP kernel's f32
    # association -- and gradients are held to 1e-3 with respect to the depth map, R and t.
    rot0, trans0 = O.get_transform_matrices(view)
    outs = []
    for r, dev, P, S in ((ro, "cpu", O.photometric_loss, O.smooth_loss), (rg, "cuda", photometric_loss, smooth_loss)):
        d = depth0.clone().to(dev).requires_grad_(True)
        r.rot_mat = rot0.clone().to(dev).requires_grad_(True)
        r.trans_xyz = trans0.clone().to(dev).requires_grad_(True)
        warped = r.warp_canon_depth(d)
        loss = P(warped[:, None], target.to(dev)[:, None]) + 0.01 * S(d)
        loss.backward()
        outs.append((warped.detach().cpu(), float(loss.detach()), d.grad.cpu(), r.rot_mat.grad.cpu(), r.trans_xyz.grad.cpu()))
    (w0, l0, g0, r0, t0), (w1, l1, g1, r1, t1) = outs
    assert w0.shape == w1.shape == (b, 64, 64)
    rel_max = lambda a, b_: float((a - b_).abs().max() / b_.abs().max())
    assert rel_max(w1, w0) < 1e-5 and abs(l1 - l0) < 1e-5 * abs(l0)
    assert rel_max(g1, g0) < 1e-3 and rel_max(r1, r0) < 1e-3 and rel_max(t1, t0) < 1e-3


def test_pt3d_demo_plumbing_silhouette_fit_decreases_loss():
    """BASELINE config 1: icosphere (80 tris), 2 views @64x64, silhouette fit; config -> renderer -> loss goes down.
    (Plumbing only: Pt3dRenderer's numerics are pytorch3d's and unpinned)"""
    from deep3dmap_amd import synthetic
    from deep3dmap_amd.config import build_renderer, load_config
    cfg = load_config(os.path.join(ROOT, "tests", "fixtures", "pt3d_like.py"))
    mv = build_renderer(cfg)
    assert mv.texture_size == 64
    v, f = synthetic.icosphere(1)
    faces = torch.from_numpy(f).cuda()
    target_v = torch.from_numpy(v * np.array([1.0, 0.7, 1.0], np.float32) * 0.8).cuda()
    with torch.no_grad():
        target = mv.silhouettes(target_v, faces, [0.0, 90.0])
    assert target.shape == (2, 64, 64) and 0.05 < float(target.mean()) < 0.9
    verts = torch.from_numpy(v * 0.8).cuda().requires_grad_(True)
    losses = mv.fit(verts, faces, target, [0.0, 90.0], steps=60, lr=0.01)
    assert np.isfinite(losses).all() and min(losses[-5:]) < 0.6 * losses[0], losses[::10]


def test_multiview_fit_gradients_against_oracle_and_graph_replay():
    """The camera-sharded fit (rank 0 of 1): loss and gradients against the oracle's renderer on the same views,
    and the HIP-graph replay against the eager step."""
    from deep3dmap_amd import synthetic
    from deep3dmap_amd.multiview import MultiViewFit
    from oracle import nr_oracle as O
    v, tri = synthetic.grid_mesh(20)
    tex = synthetic.random_textures(tri.shape[0], 2)
    eyes = synthetic.camera_ring(3)
    fit = MultiViewFit(v, tri, tex, eyes, image_size=64)
    fit.set_targets_from(synthetic.perturb(v, 0.03))
    loss, gv, gt = fit.step()
    gv, gt = gv.clone(), gt.clone()
    # oracle: same three views, same loss
    ro = O.Renderer(camera_mode="look_at", image_size=64, anti_aliasing=False)
    vt = torch.from_numpy(v).requires_grad_(True)
    tt = torch.from_numpy(tex).requires_grad_(True)
    total = 0
    targets = [t.cpu() for t in fit.targets]
    for i in range(3):
        ro.eye = [float(x) for x in eyes[i]]
        rgb, depth, alpha = ro(vt[None], torch.from_numpy(tri)[None], tt[None])
        total = total + (rgb, depth, alpha)[0].sum() * 0  # keep graph simple; per-view tensors gathered below
        if i == 0:
            R, D, A = [rgb], [depth], [alpha]
        else:
            R.append(rgb); D.append(depth); A.append(alpha)
    rgb, depth, alpha = torch.cat(R), torch.cat(D), torch.cat(A)
    mask = targets[2][:, None]
    lo = (O.photometric_loss(rgb, targets[0], mask=mask) + ((alpha - targets[2]) ** 2).sum() / (64 * 64) +
          O.photometric_loss(depth[:, None], targets[1][:, None], mask=mask))
    lo.backward()
    # (the oracle's cameras compute in the HIP kernels' f32 association: same coverage, north_star's 1e-3 on gradients)
    rel_max = lambda a, b: float((a - b).abs().max() / b.abs().max())
    assert abs(float(loss.detach()) - float(lo.detach())) < 1e-5 * abs(float(lo.detach()))
    assert rel_max(gv.cpu(), vt.grad) < 1e-3 and rel_max(gt.cpu(), tt.grad) < 1e-3
    fit.capture_graph()
    assert fit.graph_captured
    for _ in range(2):
        loss2, gv2, gt2 = fit.step()
    assert abs(float(loss2) - float(loss)) < 1e-5 * abs(float(loss)) + 1e-8
    assert _rel_l2(gv2, gv) < 1e-4 and _rel_l2(gt2, gt) < 1e-5


@pytest.mark.parametrize("S", [96, 200])
def test_large_faces_and_small_workspace_against_oracle(S):
    """Screen-filling triangles: every face is 'large' for the binner (scanned by all tiles), its bounding box
    exceeds the gathered backward's limit (atomic fallback), and with a minimal workspace the edge-gradient items
    overflow (lane-serial fallback).  All three slow paths must stay correct."""
    import ctypes
    from deep3dmap_amd import _lib
    from oracle import nr_oracle as O
    rng = np.random.default_rng(S)
    B, F = 1, 6
    xy = rng.uniform(-0.3, 0.3, (B, F, 1, 2)) + rng.uniform(-1.2, 1.2, (B, F, 3, 2))
    faces = np.concatenate([xy, rng.uniform(0.5, 3, (B, F, 3, 1))], -1).astype(np.float32)
    faces = np.concatenate([faces, faces[:, :, ::-1]], 1).copy()
    F2 = faces.shape[1]
    tex = rng.uniform(0, 1, (B, F2, 2, 2, 2, 3)).astype(np.float32)
    m = O.raster_forward(faces, tex, S, 0.1, 100.0, 1e-3, (0, 0, 0), True, True, True)
    g_rgb = rng.normal(size=(B, S, S, 3)).astype(np.float32)
    g_alpha = rng.normal(size=(B, S, S)).astype(np.float32)
    g_depth = rng.normal(size=(B, S, S)).astype(np.float32)
    gf_ref, gt_ref = O.raster_backward(m, g_rgb, g_alpha, g_depth, True, True, True)
    gf_alpha_ref, _ = O.raster_backward(m, None, g_alpha, None, False, True, False)
    L = _lib.lib()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    fd, td = dev(faces), dev(tex)
    fi = torch.full((B, S, S), -1, dtype=torch.int32, device="cuda")
    wm = torch.zeros(B, S, S, 3, device="cuda")
    dm = torch.full((B, S, S), 100.0, device="cuda")
    # forward with the MINIMUM workspace (kcap = 1: anything over one tile is a 'large' face)
    ws = torch.empty(L.d3m_forward_workspace_min_bytes(B, F2, S), dtype=torch.uint8, device="cuda")
    rc = L.d3m_forward_face_index_map(_lib.ptr(fd), _lib.ptr(fi), _lib.ptr(wm), _lib.ptr(dm), None, None, B, F2, S,
                                      0.1, 100.0, 1, 1, 1, _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    assert np.array_equal(fi.cpu().numpy(), m["face_index_map"])
    assert np.array_equal(wm.cpu().numpy(), m["weight_map"]) and np.array_equal(dm.cpu().numpy(), m["depth_map"])
    # a workspace that is too small is refused, not overrun
    assert L.d3m_forward_face_index_map(_lib.ptr(fd), _lib.ptr(fi), _lib.ptr(wm), _lib.ptr(dm), None, None, B, F2, S,
                                        0.1, 100.0, 1, 1, 1, _lib.ptr(ws), 1024, _lib.stream_ptr()) == 2
    # edge gradient: (a) the minimum workspace -> no crossing gets a record, every one is walked by the overflow kernel,
    # (b) room for a few dozen -> most overflow, (c) a roomy one -> everything goes through the line kernel; all three
    # must match the oracle.  One byte less than the minimum is refused.
    from deep3dmap_amd.neural_renderer import rasterize_ops as ops
    rgb_d, alpha_d, g_rgb_d, g_alpha_d, g_depth_d = (dev(x) for x in (m["rgb_map"], m["alpha_map"], g_rgb, g_alpha, g_depth))
    base = L.d3m_backward_pixel_map_workspace_min_bytes(B, F2, S)
    grads = []
    for room in (0, 180, 20000):            # 52 B per crossing (record + result slots + position)
        ws2 = torch.empty(base + room * 52, dtype=torch.uint8, device="cuda")
        gf = torch.zeros_like(fd)
        rc = L.d3m_backward_pixel_map(_lib.ptr(fd), _lib.ptr(fi), _lib.ptr(rgb_d), _lib.ptr(alpha_d), _lib.ptr(g_rgb_d),
                                      _lib.ptr(g_alpha_d), _lib.ptr(gf), B, F2, S, 1e-3, 1, 1, _lib.ptr(ws2), ws2.numel(),
                                      None, None, None, 0, None, _lib.stream_ptr())
        assert rc == 0
        torch.cuda.synchronize()
        grads.append(gf)
        # ... and alpha only, where the walks read grad_alpha_map / face_index_map directly (no per-pixel records): the overflow
        # kernel and the line kernels through that form too
        ga = torch.zeros_like(fd)
        assert L.d3m_backward_pixel_map(_lib.ptr(fd), _lib.ptr(fi), None, _lib.ptr(alpha_d), None, _lib.ptr(g_alpha_d),
                                        _lib.ptr(ga), B, F2, S, 1e-3, 0, 1, _lib.ptr(ws2), ws2.numel(), None, None, None, 0,
                                        None, _lib.stream_ptr()) == 0
        assert np.abs(ga.cpu().numpy() - gf_alpha_ref).max() <= 1e-3 * max(1.0, float(np.abs(gf_alpha_ref).max())), room
    assert L.d3m_backward_pixel_map(_lib.ptr(fd), _lib.ptr(fi), _lib.ptr(rgb_d), _lib.ptr(alpha_d), _lib.ptr(g_rgb_d),
                                    _lib.ptr(g_alpha_d), _lib.ptr(grads[0]), B, F2, S, 1e-3, 1, 1, _lib.ptr(ws2), base - 1,
                                    None, None, None, 0, None, _lib.stream_ptr()) == 2
    scale = max(1.0, float(np.abs(gf_ref).max()))
    # K5 / K6: bounding boxes over the gathered form's limit -> atomic fallback inside it
    gt = torch.zeros_like(td)
    ops.backward_textures(fi, dev(m["sampling_weight_map"]), dev(m["sampling_index_map"]), g_rgb_d, gt, F2, faces=fd)
    for g in grads:
        ops.backward_depth_map(fd, dm, fi, torch.zeros(1, device="cuda"), wm, g_depth_d, g, S)
        assert np.abs(g.cpu().numpy() - gf_ref).max() <= 1e-3 * scale
    assert np.abs(gt.cpu().numpy() - gt_ref).max() <= 1e-3 * max(1.0, float(np.abs(gt_ref).max()))


def test_full_size_workload_properties():
    """BASELINE headline size (100,352 triangles, 512x512, 2 views): properties that need no oracle.
    (1) every covered pixel's weights are in [0,1] and sum to 1, depth inside (near, far), face index valid and
    front-facing; (2) rendering is idempotent (bit-identical twice); (3) fill_back symmetry: reversing every
    triangle's winding renders the same silhouette; (4) per-view independence: batch of 2 == two batches of 1."""
    from deep3dmap_amd import neural_renderer as nr, synthetic
    v, tri = synthetic.grid_mesh(225)
    eyes = torch.from_numpy(synthetic.camera_ring(8)[[1, 5]]).cuda()
    vt = torch.from_numpy(v).cuda()[None].expand(2, -1, -1).contiguous()
    ft = torch.from_numpy(tri).cuda()[None].expand(2, -1, -1).contiguous()
    r = nr.Renderer(image_size=512, anti_aliasing=False, camera_mode="look_at")
    r.eye = eyes
    proj = nr.look_at(vt, eyes, _perspective_angle=30)
    from deep3dmap_amd.neural_renderer.mesh_ops import gather_faces
    faces = gather_faces(proj, ft, True)
    out = nr.Rasterize(512, 0.1, 100.0, 1e-3, [0, 0, 0], False, True, True)
    _, alpha, depth = out(faces, None)
    from deep3dmap_amd.neural_renderer.rasterize import _raster_forward
    m, _ = _raster_forward(faces, None, 512, 0.1, 100.0, 1e-3, None, False, True, True, False)
    fi, wm, dm = m["face_index_map"], m["weight_map"], m["depth_map"]
    cov = fi >= 0
    assert 0.2 < float(cov.float().mean()) < 0.9
    w = wm[cov]
    assert float(w.min()) >= 0 and float(w.max()) <= 1 and float((w.sum(-1) - 1).abs().max()) < 1e-5
    assert float(dm[cov].min()) > 0.1 and float(dm[cov].max()) < 100.0 and float(dm[~cov].min()) == 100.0
    assert int(fi.max()) < faces.shape[1]
    # winners are front-facing (KCU:111)
    idx = fi[cov].long() + (torch.arange(2, device="cuda")[:, None, None] * faces.shape[1]).expand_as(fi)[cov]
    f9 = faces.reshape(-1, 9)[idx]
    assert bool(((f9[:, 7] - f9[:, 1]) * (f9[:, 3] - f9[:, 0]) >= (f9[:, 4] - f9[:, 1]) * (f9[:, 6] - f9[:, 0])).all())
    m2, _ = _raster_forward(faces, None, 512, 0.1, 100.0, 1e-3, None, False, True, True, False)
    assert torch.equal(m2["face_index_map"], fi) and torch.equal(m2["weight_map"], wm) and torch.equal(m2["depth_map"], dm)
    sil = r(vt, ft, mode="silhouettes")
    sil_rev = r(vt, ft.flip(-1).contiguous(), mode="silhouettes")
    assert torch.equal(sil, sil_rev)
    r1 = nr.Renderer(image_size=512, anti_aliasing=False, camera_mode="look_at")
    for i in range(2):
        r1.eye = eyes[i:i + 1]
        assert torch.equal(r1(vt[i:i + 1], ft[i:i + 1], mode="silhouettes")[0], sil[i])


@pytest.mark.parametrize("S,mesh_n,big,views", [(512, 40, False, 9), (500, 160, True, 9), (1024, 60, False, 3)])
def test_one_wave_per_tile_path_equals_four_waves_per_tile_path(S, mesh_n, big, views):
    """The per-tile-list form of coverage runs ONE wave per one or two tiles when a launch has more than 32768 tiles
    (bench.py's 32 views: k_raster_tiles<..., 1, true>, the streaming form: a wave requests its second tile's list entry and
    faces while it works on its first) and FOUR waves per tile below that.  Nine views (36864 / 35721 tiles) rendered in
    one batch with the lists forced (d3m_set_coverage_form(0): "auto" would cover so small a batch by bidding) must give
    bit-identical maps to the same views rendered one by one (four waves per tile), and both to the bidding form of the
    same batch; which kernels ran is read from the launch record, not inferred.  Second case: a raster that is no multiple
    of the tile, a mesh dense enough for tiles with more than one chunk of 64 listed faces, and a screen-filling face in
    every view (the big-face list).  Third case: three views of 16384 tiles each -- an XCD's share of the tiles is less than one
    view."""
    from conftest import assert_coverage_form_ran, kernels_launched
    from deep3dmap_amd import _lib, neural_renderer as nr, synthetic
    from deep3dmap_amd.neural_renderer.mesh_ops import gather_faces
    from deep3dmap_amd.neural_renderer.rasterize import _raster_forward
    v, tri = synthetic.grid_mesh(mesh_n)
    eyes = torch.from_numpy(synthetic.camera_ring(views)).cuda()
    vt = torch.from_numpy(v).cuda()[None].expand(views, -1, -1).contiguous()
    ft = torch.from_numpy(tri).cuda()[None].expand(views, -1, -1).contiguous()
    faces = gather_faces(nr.look_at(vt, eyes, _perspective_angle=30), ft, True)
    if big:         # behind everything else, covering the whole raster: listed by every tile of its view
        wall = torch.tensor([[-3.0, -3.0, 50.0], [3.0, -3.0, 50.0], [0.0, 4.0, 50.0]], device="cuda")
        faces = torch.cat([faces, wall[None, None].expand(views, 1, 3, 3)], dim=1).contiguous()
    tex = torch.rand(views, faces.shape[1], 2, 2, 2, 3, device="cuda")
    with _lib.coverage_form("binned"), kernels_launched() as k:
        many, _ = _raster_forward(faces, tex, S, 0.1, 100.0, 1e-3, None, True, True, True, False)
    assert_coverage_form_ran(k.names, "binned")
    assert float((many["face_index_map"] >= 0).float().mean()) > 0.05
    if big:
        assert int((many["face_index_map"] == faces.shape[1] - 1).sum()) > views * S * S // 4
    with _lib.coverage_form("bidding"), kernels_launched() as k:
        bid, _ = _raster_forward(faces, tex, S, 0.1, 100.0, 1e-3, None, True, True, True, False)
    assert_coverage_form_ran(k.names, "bidding")
    for key in ("face_index_map", "weight_map", "depth_map"):
        assert torch.equal(many[key], bid[key]), key
    with _lib.coverage_form("binned"), kernels_launched() as k:
        for i in range(views):
            one, _ = _raster_forward(faces[i:i + 1].contiguous(), tex[i:i + 1].contiguous(), S, 0.1, 100.0, 1e-3, None, True,
                                     True, True, False)
            for key in ("face_index_map", "weight_map", "depth_map"):
                assert torch.equal(many[key][i], one[key][0]), (i, key)
    assert_coverage_form_ran(k.names, "binned")


def test_full_size_fit_step_fused_objective_equals_materialised_images():
    """BASELINE headline size (100,352 triangles, 512x512, 4 views): the step bench.py times -- objective evaluated
    inside the rendering node, unscaled gradient maps, no output images -- against the same step with the images
    materialised and multiview_fit_loss evaluated on them: same loss, same vertex and texture gradients (they differ
    only in where the scalar factors of the loss gradient are multiplied in)."""
    from deep3dmap_amd import synthetic
    from deep3dmap_amd.multiview import MultiViewFit
    v, tri = synthetic.grid_mesh(225)
    tex = synthetic.random_textures(tri.shape[0], 2)
    eyes = synthetic.camera_ring(32)[[0, 9, 17, 26]]
    out = []
    for inside in (True, False):
        fit = MultiViewFit(v, tri, tex, eyes, image_size=512, objective_in_renderer=inside)
        fit.set_targets_from(synthetic.perturb(v))
        loss, gv, gt = fit.step()
        assert torch.isfinite(gv).all() and torch.isfinite(gt).all()
        out.append((float(loss), gv.clone(), gt.clone()))
    (la, gva, gta), (lb, gvb, gtb) = out
    assert abs(la - lb) <= 1e-5 * abs(lb)
    assert float(gvb.abs().max()) > 0 and float(gtb.abs().max()) > 0
    assert _rel_l2(gva, gvb) < 1e-4 and _rel_l2(gta, gtb) < 1e-5


def test_graph_replay_with_host_syncs():
    """Regression: replaying the captured step with host synchronisation, read-backs and fresh allocations between
    replays (what a training loop with logging does) must keep reproducing the eager step."""
    from deep3dmap_amd import synthetic
    from deep3dmap_amd.multiview import MultiViewFit
    v, tri = synthetic.grid_mesh(40)
    fit = MultiViewFit(v, tri, synthetic.random_textures(tri.shape[0], 2), synthetic.camera_ring(4), image_size=96)
    fit.set_targets_from(synthetic.perturb(v))
    l0, gv0, gt0 = fit.step()
    gv0, gt0, l0 = gv0.clone(), gt0.clone(), float(l0)
    fit.capture_graph()
    for i in range(6):
        l, gv, gt = fit.step()
        torch.cuda.synchronize()
        assert abs(float(l) - l0) <= 1e-5 * abs(l0)
        assert _rel_l2(gv, gv0) < 1e-4 and _rel_l2(gt, gt0) < 1e-5
        junk = torch.zeros(1 << 20, device="cuda")   # allocator traffic between replays
        del junk


def test_steps_issued_on_the_fits_own_stream_need_no_fences():
    """A loop wrapped in `with torch.cuda.stream(fit.stream)` (bench.py's timed loops) skips the per-step
    fences against the caller's stream: replays issued back to back on that stream, with in-place parameter updates between
    them, reproduce what the same updates give step by step from the default stream."""
    from deep3dmap_amd import synthetic
    from deep3dmap_amd.multiview import MultiViewFit

    def run(own_stream):
        v, tri = synthetic.grid_mesh(40)
        fit = MultiViewFit(v, tri, synthetic.random_textures(tri.shape[0], 2), synthetic.camera_ring(4), image_size=96)
        fit.set_targets_from(synthetic.perturb(v))
        fit.capture_graph()
        ctx = torch.cuda.stream(fit.stream) if own_stream else torch.cuda.stream(torch.cuda.current_stream())
        losses = []
        with ctx:
            for _ in range(5):
                loss, gv, _ = fit.step()
                losses.append(loss.clone())
                with torch.no_grad():
                    fit.vertices.sub_(0.05 * gv)         # in place, on the stream the next replay follows on
        torch.cuda.synchronize()
        return [float(x) for x in losses], fit.vertices.detach().clone()

    l_own, v_own = run(True)
    l_ref, v_ref = run(False)
    assert l_own[0] != l_own[-1]                                    # the updates do change the objective
    assert np.allclose(l_own, l_ref, rtol=1e-5, atol=0) and _rel_l2(v_own, v_ref) < 1e-5


def test_graph_capture_with_retained_autograd_state():
    """NrRenderer keeps rot_mat / trans_xyz (and so the previous step's autograd graph and the view's
    AccumulateGrad node) alive between steps; capture must still work because every step runs on one stream
    (deep3dmap_amd/graph.py)."""
    from deep3dmap_amd.core import NrRenderer, photometric_loss
    from deep3dmap_amd.graph import CapturedStep
    b, hw = 4, 32
    r = NrRenderer(dict(min_depth=0.9, max_depth=1.1, rot_center_depth=1.0, fov=10), hw)
    depth = (1.0 + 0.05 * torch.rand(b, hw, hw)).cuda().requires_grad_(True)
    view = (torch.rand(b, 6) * 0.1).cuda().requires_grad_(True)
    target = torch.ones(b, 1, hw, hw, device="cuda")

    def step():
        depth.grad = None
        view.grad = None
        r.set_transform_matrices(view)
        loss = photometric_loss(r.warp_canon_depth(depth)[:, None], target)
        loss.backward()
        return loss.detach()

    run = CapturedStep(step)
    l0 = float(run())
    gd0, gv0 = depth.grad.clone(), view.grad.clone()
    depth.grad = None
    view.grad = None
    run.capture()
    for _ in range(4):
        l = run()
        torch.cuda.synchronize()
        assert abs(float(l) - l0) <= 1e-5 * abs(l0)
        assert _rel_l2(depth.grad, gd0) < 1e-4 and _rel_l2(view.grad, gv0) < 1e-4


@pytest.mark.parametrize("aa", [False, True])
@pytest.mark.parametrize("groups", [2, 3])
def test_view_groups_equal_one_pipeline(groups, aa):
    """The lit render node run as concurrent view groups (rasterize._RasterizeLit, "VIEW GROUPS"): the fused fit
    objective and its gradients (the groups are shards of it), and plain render() with per-view textures -- images
    bit for bit, gradients to rounding -- against the single pipeline."""
    from deep3dmap_amd import neural_renderer as nr, synthetic
    from deep3dmap_amd.multiview import MultiViewFit
    v, tri = synthetic.grid_mesh(30)
    tex = synthetic.random_textures(tri.shape[0], 2)
    eyes = synthetic.camera_ring(5)
    res = []
    for g in (1, groups):
        fit = MultiViewFit(v, tri, tex, eyes, image_size=96, view_groups=g, anti_aliasing=aa)    # (aa: the pooled records)
        fit.set_targets_from(synthetic.perturb(v, 0.03))
        loss, gv, gt = fit.step()
        eager = (float(loss), gv.clone(), gt.clone())
        fit.capture_graph()
        for _ in range(3):
            loss, gv, gt = fit.step()
        assert abs(float(loss) - eager[0]) <= 1e-6 * abs(eager[0]) and _rel_l2(gv, eager[1]) < 1e-5
        res.append(eager)
    (l1, gv1, gt1), (lg, gvg, gtg) = res
    assert abs(lg - l1) <= 1e-5 * abs(l1)
    assert float((gvg - gv1).abs().max()) <= 1e-4 * float(gv1.abs().max())      # float atomics meet in another order
    assert float((gtg - gt1).abs().max()) <= 1e-4 * float(gt1.abs().max())
    if aa:
        return
    # render(): per-view textures and world vertices with gradients (light gradient path), anti-aliasing on
    B = 5
    vt = torch.from_numpy(v).cuda()[None].repeat(B, 1, 1) * torch.linspace(0.9, 1.1, B, device="cuda")[:, None, None]
    ft = torch.from_numpy(tri).cuda()[None].repeat(B, 1, 1)
    tx = torch.rand(B, tri.shape[0], 2, 2, 2, 3, device="cuda")
    outs = []
    for g in (1, groups):
        r = nr.Renderer(image_size=48, anti_aliasing=True, camera_mode="look_at")
        r.eye = torch.from_numpy(eyes).cuda()
        r.view_groups = g
        vg, tg = vt.clone().requires_grad_(True), tx.clone().requires_grad_(True)
        rgb, depth, alpha = r(vg, ft, tg)
        (rgb.square().sum() + alpha.sum() + depth.clamp(max=5).sum()).backward()
        outs.append((rgb.detach(), depth.detach(), alpha.detach(), vg.grad, tg.grad))
    for a, b in zip(outs[0][:3], outs[1][:3]):
        assert torch.equal(a, b)
    for a, b in zip(outs[0][3:], outs[1][3:]):
        assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max())


def test_config5_per_gpu_shard_in_one_batch_equals_its_quarters():
    """BASELINE config 5 as ONE GPU runs it on an 8-GPU node: 32 of the 256 cameras, the 1,002,528-triangle mesh,
    1024x1024 -- 64 M (view, face) slots in one launch, the size that stresses the int-indexed capacities of the binning
    passes (kcap * B * F) and of the edge plan.  No oracle finishes at this size, so the check is a property: the 32-camera
    batch must give the loss and the vertex / texture gradients of the same cameras run as four batches of eight
    (each normalised by the whole batch's mask sum, i.e. as shards of one objective)."""
    from deep3dmap_amd import synthetic
    from deep3dmap_amd.multiview import MultiViewFit
    v, tri = synthetic.grid_mesh(709)
    assert tri.shape[0] == 1002528
    tex = synthetic.random_textures(tri.shape[0], 2)
    eyes = synthetic.camera_ring(256)[:32]
    target = synthetic.perturb(v)
    whole = MultiViewFit(v, tri, tex, eyes, image_size=1024)
    whole.set_targets_from(target)
    loss, gv, gt = whole.step()
    loss, gv, gt = float(loss), gv.clone(), gt.clone()
    assert np.isfinite(loss) and bool(torch.isfinite(gv).all()) and bool(torch.isfinite(gt).all())
    cover = float(whole.targets[2].mean())
    assert 0.2 < cover < 0.9, cover
    mask_sum = whole.mask_sum.clone()
    del whole
    torch.cuda.empty_cache()
    acc = None
    for q in range(4):
        part = MultiViewFit(v, tri, tex, eyes, image_size=1024, rank=q, world_size=4)      # (no process group: local sums)
        part.set_targets_from(target)
        part.mask_sum = mask_sum
        l, g1, g2 = part.step()
        acc = [float(l), g1.clone(), g2.clone()] if acc is None else [acc[0] + float(l), acc[1] + g1, acc[2] + g2]
        del part
        torch.cuda.empty_cache()
    assert abs(acc[0] - loss) <= 1e-5 * abs(loss), (acc[0], loss)
    assert _rel_l2(acc[1], gv) < 1e-4 and _rel_l2(acc[2], gt) < 1e-5


def test_plan_capacity_follows_the_raster_size():
    """The edge plan's default capacity goes by faces OR pixels (eg_default_crossings): BASELINE config 4's mesh at 1024^2
    -- the same 100,352 triangles, four times as large on screen, about a million crossings per view where two per face
    are 401 k -- must still get its records (k_edge_lines walks them) instead of falling back to the gather pass's own walk
    of every crossing from global memory (csrc/d3m_edge_grad.h, 4.): 8.0 ms instead of 0.3 per 8 views when the capacity
    went by faces alone."""
    from conftest import kernels_launched
    from deep3dmap_amd import synthetic
    from deep3dmap_amd.multiview import MultiViewFit
    v, tri = synthetic.grid_mesh(225)
    tex = synthetic.random_textures(tri.shape[0], 2)
    fit = MultiViewFit(v, tri, tex, synthetic.camera_ring(32)[[2, 19]], image_size=1024)
    fit.set_targets_from(synthetic.perturb(v))
    fit.step()
    with kernels_launched() as k:
        loss, gv, _ = fit.step()
    assert torch.isfinite(loss) and torch.isfinite(gv).all()
    lines, gather = k.times["k_edge_lines"][1], k.times["k_edge_gather"][1]
    assert gather < lines, (lines, gather)       # (ms; the fallback takes 30x the line kernel's time when it runs, and the line kernel then none)


@pytest.mark.parametrize("mode", ["silhouettes", "depth"])
def test_modes_over_the_indexed_mesh_at_full_size(mode):
    """The silhouette / depth modes as one node (rasterize._RasterizeMeshModes) on BASELINE config 4's mesh and raster -- eight
    of its cameras, 100,352 triangles @512x512, the binned form of coverage as at 32 views -- against the operator sequence
    of the reference (Renderer.mesh_modes off), which the reference-parity tests pin: images bit for bit; the vertex gradient
    element-wise within the 1e-3 contract (|d| <= 1e-3 |ref| + 1e-5 max|ref| for all but a 1e-5 fraction: both sides add the
    same terms with float atomics in another order)."""
    from deep3dmap_amd import _lib, neural_renderer as nr, synthetic
    v_np, tri_np = synthetic.grid_mesh(225)
    eyes = torch.from_numpy(synthetic.camera_ring(32)[::4].copy()).float().cuda()
    tri = torch.from_numpy(tri_np).int().cuda()[None]
    gen = torch.Generator(device="cuda").manual_seed(2)
    res = []
    with _lib.coverage_form("binned"):
        for on in (True, False):
            r = nr.Renderer(image_size=512, anti_aliasing=False, camera_mode="look_at", fill_back=True)
            r.eye, r.mesh_modes = eyes, on
            v = torch.from_numpy(v_np).float().cuda()[None].requires_grad_(True)
            image = getattr(r, "render_" + mode)(v, tri)
            if on:
                w = torch.rand(image.shape, device="cuda", generator=gen) - 0.4
            (image * w).sum().backward()
            res.append((image.detach(), v.grad.clone()))
    assert torch.equal(res[0][0], res[1][0])
    g, ref = res[0][1], res[1][1]
    scale = float(ref.abs().max())
    bad = ((g - ref).abs() > 1e-3 * ref.abs() + 1e-5 * scale).float().mean()
    assert scale > 0 and float(bad) <= 1e-5, float(bad)
    assert float((g - ref).abs().max()) <= 1e-4 * scale
