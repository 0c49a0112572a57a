"""GPU parity tests of the five drop-in operators (section A of include/d3m_raster.h), called through
the C ABI, against the committed golden vectors (= the reference's kernels) and the oracle."""
import numpy as np
import pytest
import torch

from conftest import assert_coverage_form_ran, golden_case, kernels_launched

pytestmark = pytest.mark.gpu

CASES = ["rand_b2_f24_s32_ts2", "rand_b1_f64_s64_ts4", "rand_b3_f12_s16_ts1", "nearfar_b1_f32_s32",
         "ties_zero_batch_b2_f32_s32", "fillback_small_b2_f80_s48", "grid_on_pixel_centres_b1_f128_s32"]

# Tolerances.  north_star: 1e-4 on depth/barycentrics, 1e-3 on backward gradients.  The forward is
# built to be bit-identical (same f32 operations, no FMA contraction), so it is held to exact equality
# on indices and 1e-6 on floats; gradients are float-atomic / re-associated sums.
FWD_ATOL = 1e-6
GRAD_RTOL = 1e-3


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _grad_close(got, ref):
    """north_star: 1e-3 on backward gradients -- of the tensor's own scale (its largest entry), no absolute floor"""
    scale = float(np.abs(ref).max())
    if scale == 0.0:
        return not np.any(got)
    return np.abs(got - ref).max() <= GRAD_RTOL * scale


def _forward(c):
    from deep3dmap_amd.neural_renderer import rasterize_ops as ops
    faces, tex = _dev(c["faces"]), _dev(c["textures"])
    B, F = faces.shape[:2]
    S = int(c["image_size"])
    fi = torch.full((B, S, S), -1, dtype=torch.int32, device="cuda")
    wm = torch.zeros(B, S, S, 3, device="cuda")
    dm = torch.full((B, S, S), float(c["far"]), device="cuda")
    fim = torch.zeros(B, S, S, 3, 3, device="cuda")
    finv = torch.zeros_like(faces)
    ops.forward_face_index_map(faces, fi, wm, dm, fim, finv, S, float(c["near"]), float(c["far"]), 1, 1, 1)
    rgb = torch.zeros(B, S, S, 3, device="cuda")
    si = torch.zeros(B, S, S, 8, dtype=torch.int32, device="cuda")
    sw = torch.zeros(B, S, S, 8, device="cuda")
    ops.forward_texture_sampling(faces, tex, fi, wm, dm, rgb, si, sw, S, float(c["eps"]))
    return dict(faces=faces, textures=tex, face_index_map=fi, weight_map=wm, depth_map=dm, face_inv_map=fim,
                faces_inv=finv, rgb_sampled=rgb, sampling_index_map=si, sampling_weight_map=sw)


@pytest.mark.parametrize("name", CASES)
def test_forward_ops_match_reference_vectors(golden, name, coverage):
    c = golden_case(golden, name)
    with kernels_launched() as k:
        m = _forward(c)
    assert_coverage_form_ran(k.names, coverage)
    assert np.array_equal(m["face_index_map"].cpu().numpy(), c["face_index_map"])
    assert np.array_equal(m["sampling_index_map"].cpu().numpy(), c["sampling_index_map"])
    for key in ("weight_map", "depth_map", "face_inv_map", "faces_inv", "sampling_weight_map"):
        got, ref = m[key].cpu().numpy(), c[key]
        assert np.allclose(got, ref, rtol=0, atol=FWD_ATOL, equal_nan=True), (key, np.nanmax(np.abs(got - ref)))
    # the golden rgb_map is after the background blend; compare covered pixels only
    cov = c["face_index_map"] >= 0
    assert np.allclose(m["rgb_sampled"].cpu().numpy()[cov], c["rgb_map"][cov], rtol=0, atol=FWD_ATOL)


@pytest.mark.parametrize("name", CASES)
def test_backward_ops_match_reference_vectors(golden, name):
    from deep3dmap_amd.neural_renderer import rasterize_ops as ops
    c = golden_case(golden, name)
    m = _forward(c)
    S = int(c["image_size"])
    rgb, alpha = _dev(c["rgb_map"]), _dev(c["alpha_map"])
    g_rgb, g_alpha, g_depth = _dev(c["grad_rgb_map"]), _dev(c["grad_alpha_map"]), _dev(c["grad_depth_map"])
    eps = float(c["eps"])
    # K4 alone, rgb + alpha
    gf = torch.zeros_like(m["faces"])
    ops.backward_pixel_map(m["faces"], m["face_index_map"], rgb, alpha, g_rgb, g_alpha, gf, S, eps, 1, 1)
    assert _grad_close(gf.cpu().numpy(), c["grad_faces_pixel_rgba"])
    # K4 alone, alpha only (rgb tensors are 1-element dummies as in rasterize.py:59)
    gf = torch.zeros_like(m["faces"])
    dummy = torch.zeros(1, device="cuda")
    ops.backward_pixel_map(m["faces"], m["face_index_map"], dummy, alpha, dummy, g_alpha, gf, S, eps, 0, 1)
    assert _grad_close(gf.cpu().numpy(), c["grad_faces_pixel_alpha"])
    # K6 alone, with the stored face_inv_map and with it recomputed
    for fim in (m["face_inv_map"], dummy):
        gf = torch.zeros_like(m["faces"])
        ops.backward_depth_map(m["faces"], m["depth_map"], m["face_index_map"], fim, m["weight_map"], g_depth, gf, S)
        assert _grad_close(gf.cpu().numpy(), c["grad_faces_depth"])
    # composed order K4 -> K5 -> K6 (rasterize.py:141-151)
    gf = torch.zeros_like(m["faces"])
    gt = torch.zeros_like(m["textures"])
    ops.backward_pixel_map(m["faces"], m["face_index_map"], rgb, alpha, g_rgb, g_alpha, gf, S, eps, 1, 1)
    ops.backward_textures(m["face_index_map"], m["sampling_weight_map"], m["sampling_index_map"], g_rgb, gt,
                          m["faces"].shape[1])
    # the same sums gathered per visible face (optional `faces` argument) instead of scattered atomically
    gt2 = torch.zeros_like(m["textures"])
    ops.backward_textures(m["face_index_map"], m["sampling_weight_map"], m["sampling_index_map"], g_rgb, gt2,
                          m["faces"].shape[1], faces=m["faces"])
    assert _grad_close(gt2.cpu().numpy(), c["grad_textures"])
    ops.backward_depth_map(m["faces"], m["depth_map"], m["face_index_map"], m["face_inv_map"], m["weight_map"],
                           g_depth, gf, S)
    assert _grad_close(gf.cpu().numpy(), c["grad_faces_all"])
    assert _grad_close(gt.cpu().numpy(), c["grad_textures"])


def test_ops_reject_cpu_and_noncontiguous_tensors():
    """rasterize_cuda.cpp:66-68: CHECK_CUDA / CHECK_CONTIGUOUS raise RuntimeError."""
    from deep3dmap_amd.neural_renderer import rasterize_ops as ops
    f = torch.zeros(1, 2, 3, 3)
    with pytest.raises(RuntimeError):
        ops.forward_face_index_map(f, f, f, f, f, f, 8, 0.1, 100.0, 0, 0, 0)
    fc = torch.zeros(1, 2, 3, 6, device="cuda")[..., ::2]
    with pytest.raises(RuntimeError):
        ops.forward_face_index_map(fc, fc, fc, fc, fc, fc, 8, 0.1, 100.0, 0, 0, 0)


@pytest.mark.parametrize("S,F,size", [(64, 500, 0.05), (100, 300, 0.3), (37, 64, 1.5)])
def test_forward_equals_oracle_bruteforce_on_random_scenes(S, F, size, coverage):
    """Both forms of coverage (per-tile lists | bidding) vs the oracle's brute-force loop: tiny, medium and
    screen-filling triangles, image sizes that are not multiples of the 8-pixel tile."""
    from deep3dmap_amd.neural_renderer import rasterize_ops as ops
    from oracle import nr_oracle as O
    rng = np.random.default_rng(S * 1000 + F)
    B = 2
    xy = rng.uniform(-1.2, 1.2, (B, F, 1, 2)) + rng.uniform(-size, size, (B, F, 3, 2))
    faces = np.concatenate([xy, rng.uniform(0.3, 4.0, (B, F, 3, 1))], -1).astype(np.float32)
    faces = np.concatenate([faces, faces[:, :, ::-1]], 1).copy()             # fill_back pairs
    ref = O.raster_forward(faces, None, S, 0.5, 3.5, 1e-3, None, False, True, True)
    fd = _dev(faces)
    fi = torch.full((B, S, S), -1, dtype=torch.int32, device="cuda")
    wm = torch.zeros(B, S, S, 3, device="cuda")
    dm = torch.full((B, S, S), 3.5, device="cuda")
    fim = torch.zeros(B, S, S, 3, 3, device="cuda")
    with kernels_launched() as k:
        ops.forward_face_index_map(fd, fi, wm, dm, fim, torch.zeros_like(fd), S, 0.5, 3.5, 0, 1, 1)
    assert_coverage_form_ran(k.names, coverage)
    assert np.array_equal(fi.cpu().numpy(), ref["face_index_map"])
    assert np.array_equal(wm.cpu().numpy(), ref["weight_map"])
    assert np.array_equal(dm.cpu().numpy(), ref["depth_map"])
    assert np.array_equal(fim.cpu().numpy(), ref["face_inv_map"])


@pytest.mark.parametrize("masked", [False, True])
def test_edge_gradient_crowded_lines_and_clipped_walks(masked):
    """K4 on a scene built to stress the line kernel: 700 flat 40 x 1.4-pixel faces stacked down a 1024-pixel image,
    so every column line under them carries > 1024 walk segments (two length-sort chunks), short and long ones mixed;
    with `masked` the gradient maps vanish outside a band, so walks are clipped to the non-zero extent (and many
    disappear).  Against the oracle's per-face walk (KCU:245-503)."""
    from deep3dmap_amd.neural_renderer import rasterize_ops as ops
    from oracle import nr_oracle as O
    S, n = 1024, 700
    rng = np.random.default_rng(7 + int(masked))
    y0 = np.linspace(-0.95, 0.93, n)
    x0 = rng.uniform(-0.05, -0.03, n)             # every face crosses the columns around x = 0
    h, w = 1.4 * 2 / S, 40 * 2 / S
    # counter-clockwise in NDC so that they are front-facing; slightly tilted so edges cross rows too
    tri = np.stack([np.stack([x0, y0], -1), np.stack([x0 + w, y0 + 0.3 * h], -1), np.stack([x0 + 0.4 * w, y0 + h], -1)], 1)
    z = rng.uniform(1.0, 2.0, (n, 3, 1))
    faces = np.concatenate([tri, z], -1)[None].astype(np.float32)
    faces = np.concatenate([faces, faces[:, :, ::-1]], 1).copy()
    m = O.raster_forward(faces, rng.uniform(0, 1, (1, 2 * n, 2, 2, 2, 3)).astype(np.float32), S, 0.1, 100.0, 1e-3,
                         (0.1, 0.2, 0.3), True, True, False)
    assert (m["face_index_map"] >= 0).sum() > 10000
    assert ((m["face_index_map"][0, :, S // 2] >= 0).sum()) > 300     # the central column is owned by hundreds of faces
    g_rgb = rng.normal(size=(1, S, S, 3)).astype(np.float32)
    g_alpha = rng.normal(size=(1, S, S)).astype(np.float32)
    if masked:
        band = np.zeros((1, S, S), np.float32)
        band[:, 100:900, 380:520] = 1.0
        g_rgb *= band[..., None]
        g_alpha *= band
    gf_ref, _ = O.raster_backward(m, g_rgb, g_alpha, None, True, True, False)
    fd = _dev(faces)
    gf = torch.zeros_like(fd)
    ops.backward_pixel_map(fd, _dev(m["face_index_map"]), _dev(m["rgb_map"]), _dev(m["alpha_map"]), _dev(g_rgb), _dev(g_alpha),
                           gf, S, 1e-3, True, True)
    assert np.abs(gf_ref).max() > 0 and _grad_close(gf.cpu().numpy(), gf_ref)


def test_edge_gradient_pass_with_more_long_segments_than_the_queue_holds():
    """K4 on rows that carry more LONG segments than one pass of the line kernel can queue: 250 shingled triangles, each
    8 pixels wide at its base and 2 pixels to the right of (and nearer than) the one before, so that on every row near the
    base a face's left edge gives a record with a long outward AND a long inward walk and its right edge one with a long
    inward walk: ~500 records, ~750 queue items in a pass of 512 records against a queue of 512 -- the pass is re-taken by
    halves (k_edge_lines, "pass_total > EG_QUEUE").  Against the oracle's per-face walk (KCU:245-503)."""
    from deep3dmap_amd.neural_renderer import rasterize_ops as ops
    from oracle import nr_oracle as O
    S, n = 512, 250
    px = 2.0 / S
    x0 = -1.0 + 2.5 * px + 2 * px * np.arange(n)
    y0, y1 = -0.5, 0.62                                                   # tall: the rows near the base see ~8 px of width
    z = 2.0 - 0.003 * np.arange(n)                                        # every face nearer than the one to its left
    tri = np.stack([np.stack([x0, np.full(n, y0)], -1), np.stack([x0 + 8.2 * px, np.full(n, y0 + 0.3 * px)], -1),
                    np.stack([x0 + 4.1 * px, np.full(n, y1)], -1)], 1)    # counter-clockwise: front-facing
    faces = np.concatenate([tri, np.broadcast_to(z[:, None, None], (n, 3, 1))], -1)[None].astype(np.float32)
    faces = np.concatenate([faces, faces[:, :, ::-1]], 1).copy()
    rng = np.random.default_rng(11)
    m = O.raster_forward(faces, rng.uniform(0, 1, (1, 2 * n, 2, 2, 2, 3)).astype(np.float32), S, 0.1, 100.0, 1e-3,
                         (0.1, 0.2, 0.3), True, True, False)
    row = m["face_index_map"][0, S // 4 + 8]                              # a row a few pixels above the base
    assert len(np.unique(row[row >= 0])) > 200                            # ... is shared by more than 200 owners
    g_rgb = rng.normal(size=(1, S, S, 3)).astype(np.float32)
    g_alpha = rng.normal(size=(1, S, S)).astype(np.float32)
    gf_ref, _ = O.raster_backward(m, g_rgb, g_alpha, None, True, True, False)
    fd = _dev(faces)
    gf = torch.zeros_like(fd)
    ops.backward_pixel_map(fd, _dev(m["face_index_map"]), _dev(m["rgb_map"]), _dev(m["alpha_map"]), _dev(g_rgb), _dev(g_alpha),
                           gf, S, 1e-3, True, True)
    assert np.abs(gf_ref).max() > 0 and _grad_close(gf.cpu().numpy(), gf_ref)


@pytest.mark.parametrize("gradient", ["grow_band", "dense", "single_pixels", "nan"])
def test_alpha_only_edge_gradient_sparse_and_dense_walks(gradient, monkeypatch):
    """K4 with return_rgb == 0 (render_silhouettes): the outward walks' terms are (alpha(d) - 1) * grad_alpha(d), non-zero
    only at uncovered pixels with a negative gradient (and the inward walks' only at covered ones with a positive one), so
    lines of few such pixels are walked by k_edge_lines_alpha (two waves per line over sorted lists of them) and, beyond its
    lists, by k_edge_lines' own sparse form (SPARSE OUTWARD WALKS) -- unless the line holds more of them than
    D3M_EG_SPARSE_MAX (default S / 4; 0 = neither), which walks densely as the rgb modes do.  Both forms against the oracle's per-face walk
    (KCU:245-503), and against each other far below the tolerance: a thin band of negative gradients outside the
    silhouette (the shape a silhouette fit produces), dense random gradients (every line over the limit at the default;
    forced sparse with the limit raised), isolated pixels, and a NaN gradient (kept by `diff_grad <= 0` being false,
    KCU:401: it must poison the same faces in both forms)."""
    from deep3dmap_amd.neural_renderer import rasterize_ops as ops
    from oracle import nr_oracle as O
    S, n = 192, 300
    rng = np.random.default_rng(77)
    xy = rng.uniform(-0.55, 0.55, (2, n, 1, 2)) + rng.uniform(-1, 1, (2, n, 3, 2)) * rng.choice([0.02, 0.08, 0.3], (2, n, 1, 1))
    faces = np.concatenate([xy, rng.uniform(1.0, 2.0, (2, n, 3, 1))], -1).astype(np.float32)
    faces = np.concatenate([faces, faces[:, :, ::-1]], 1).copy()
    m = O.raster_forward(faces, None, S, 0.1, 100.0, 1e-3, None, False, True, False)
    alpha = m["alpha_map"]
    assert 0.15 < alpha.mean() < 0.85
    if gradient == "grow_band":         # 2 (alpha - target) / P with a target that is the silhouette dilated by ~6 pixels
        import scipy.ndimage as ndi
        target = np.stack([ndi.binary_dilation(a > 0, iterations=6) for a in alpha]).astype(np.float32)
        target[:, :, : S // 3] = alpha[:, :, : S // 3] * 0.0      # ... and shrunk away on the left third (positive gradients)
        g_alpha = (2.0 * (alpha - target) / (S * S)).astype(np.float32)
    elif gradient == "dense":
        g_alpha = rng.normal(size=alpha.shape).astype(np.float32)
    else:
        g_alpha = np.zeros_like(alpha)
        ys, xs = rng.integers(0, S, 40), rng.integers(0, S, 40)
        g_alpha[rng.integers(0, 2, 40), ys, xs] = -rng.uniform(0.5, 2.0, 40).astype(np.float32)
        if gradient == "nan":
            unc = np.argwhere(alpha[0] == 0)
            y, x = unc[len(unc) // 2]
            g_alpha[0, y, x] = np.nan
    gf_ref, _ = O.raster_backward(m, None, g_alpha, None, False, True, False)
    fd, fi, am, ga = _dev(faces), _dev(m["face_index_map"]), _dev(alpha), _dev(g_alpha)
    dummy = torch.zeros(1, device="cuda")
    got = {}
    for form, limit in (("default", None), ("dense", "0"), ("sparse", str(S)), ("packed", None)):
        # ("packed": D3M_EG_DIRECT=0 -- the gradients as per-pixel records with line extents, the form the rgb modes and
        # the fused objectives use; the default reads grad_alpha_map / face_index_map directly, d3m_edge_grad.h "DIRECT")
        monkeypatch.setenv("D3M_EG_DIRECT", "0" if form == "packed" else "1")
        if limit is None:
            monkeypatch.delenv("D3M_EG_SPARSE_MAX", raising=False)
        else:
            monkeypatch.setenv("D3M_EG_SPARSE_MAX", limit)
        gf = torch.zeros_like(fd)
        ops.backward_pixel_map(fd, fi, dummy, am, dummy, ga, gf, S, 1e-3, False, True)
        got[form] = gf.cpu().numpy()
    if gradient == "nan":
        poisoned = np.isnan(gf_ref).any(axis=(2, 3))
        assert poisoned.any()
        for form, g in got.items():
            assert np.array_equal(np.isnan(g).any(axis=(2, 3)), poisoned), form
            ok = ~np.isnan(gf_ref)
            assert np.abs(g[ok] - gf_ref[ok]).max() <= GRAD_RTOL * np.abs(gf_ref[ok]).max(), form
        return
    assert np.abs(gf_ref).max() > 0
    for form, g in got.items():
        assert _grad_close(g, gf_ref), (form, np.abs(g - gf_ref).max() / np.abs(gf_ref).max())
    scale = np.abs(gf_ref).max()
    assert np.abs(got["sparse"] - got["dense"]).max() <= 1e-4 * scale
    # ... and no further from the reference than the dense walk (the same v_rcp_f32 quotients, fewer terms)
    assert np.abs(got["sparse"] - gf_ref).max() <= np.abs(got["dense"] - gf_ref).max() + 2e-5 * scale


def test_edge_gradient_on_an_image_wider_than_the_line_window():
    """K4 at S = 2304: above 2048 pixels per line the plan's count pass merges its line counters by key instead of in
    the LDS line window, and the line kernel clamps instead of padding -- small and large faces, two views, against
    the oracle's per-face walk."""
    from deep3dmap_amd.neural_renderer import rasterize_ops as ops
    from oracle import nr_oracle as O
    S, n = 2304, 48
    rng = np.random.default_rng(23)
    xy = rng.uniform(-0.9, 0.9, (2, n, 1, 2)) + rng.uniform(-1, 1, (2, n, 3, 2)) * rng.choice([0.004, 0.05, 0.4], (2, n, 1, 1))
    faces = np.concatenate([xy, rng.uniform(1.0, 2.0, (2, n, 3, 1))], -1).astype(np.float32)
    faces = np.concatenate([faces, faces[:, :, ::-1]], 1).copy()
    m = O.raster_forward(faces, rng.uniform(0, 1, (2, 2 * n, 2, 2, 2, 3)).astype(np.float32), S, 0.1, 100.0, 1e-3,
                         (0.1, 0.2, 0.3), True, True, False)
    assert (m["face_index_map"] >= 0).sum() > 20000
    band = np.zeros((2, S, S), np.float32)
    band[:, 300:2100, 200:2000] = 1.0
    g_rgb = rng.normal(size=(2, S, S, 3)).astype(np.float32) * band[..., None]
    g_alpha = rng.normal(size=(2, S, S)).astype(np.float32) * band
    gf_ref, _ = O.raster_backward(m, g_rgb, g_alpha, None, True, True, False)
    fd = _dev(faces)
    gf = torch.zeros_like(fd)
    ops.backward_pixel_map(fd, _dev(m["face_index_map"]), _dev(m["rgb_map"]), _dev(m["alpha_map"]), _dev(g_rgb), _dev(g_alpha),
                           gf, S, 1e-3, True, True)
    assert np.abs(gf_ref).max() > 0 and _grad_close(gf.cpu().numpy(), gf_ref)


@pytest.mark.parametrize("seed", range(60))
def test_backward_operators_differential_fuzz(seed, coverage):
    """Random small scenes through K4 / K5 / K6 against the oracle: triangle sizes from sub-pixel to screen-filling,
    vertices snapped to pixel centres or pixel edges in some scenes (integer crossings -> the t == 0 pixel of inward
    walks, shared-edge ties), duplicated and degenerate faces, faces partly or wholly off screen, and gradient maps
    that are dense, masked to a box (clipped walks) or a single pixel."""
    from deep3dmap_amd.neural_renderer import rasterize_ops as ops
    from oracle import nr_oracle as O
    rng = np.random.default_rng(1000 + seed)
    B = int(rng.integers(1, 3))
    S = int(rng.choice([16, 24, 33, 48, 64]))
    Fn = int(rng.integers(1, 40))
    size = float(rng.choice([0.03, 0.15, 0.6, 2.0]))
    xy = rng.uniform(-1.2, 1.2, (B, Fn, 1, 2)) + rng.uniform(-size, size, (B, Fn, 3, 2))
    mode = seed % 4
    if mode == 1:                                  # pixel centres: (2i + 1 - S) / S
        xy = (2 * np.round((xy * S + S - 1) / 2) + 1 - S) / S
    elif mode == 2:                                # pixel edges
        xy = np.round(xy * S / 2) * 2 / S
    faces = np.concatenate([xy, rng.uniform(0.3, 4.0, (B, Fn, 3, 1))], -1).astype(np.float32)
    if Fn > 3:
        faces[:, -1] = faces[:, 0]                 # exact duplicate: equal depth, lowest index wins
        faces[:, -2, 1] = faces[:, -2, 0]          # two coincident vertices: zero area
    faces = np.concatenate([faces, faces[:, :, ::-1]], 1).copy()
    F2, ts = faces.shape[1], 2
    tex = rng.uniform(0, 1, (B, F2, ts, ts, ts, 3)).astype(np.float32)
    m = O.raster_forward(faces, tex, S, 0.5, 3.5, 1e-3, (0.1, 0.2, 0.3), True, True, True)
    g_rgb = rng.normal(size=(B, S, S, 3)).astype(np.float32)
    g_alpha = rng.normal(size=(B, S, S)).astype(np.float32)
    g_depth = rng.normal(size=(B, S, S)).astype(np.float32)
    gmode = (seed // 4) % 3
    if gmode == 1:
        box = np.zeros((B, S, S), np.float32)
        box[:, S // 4: 3 * S // 4, S // 3: S - 2] = 1
        g_rgb, g_alpha = g_rgb * box[..., None], g_alpha * box
    elif gmode == 2:
        one = np.zeros((B, S, S), np.float32)
        one[:, S // 2, S // 2] = 1
        g_rgb, g_alpha = g_rgb * 0, g_alpha * one
    gf_ref, gt_ref = O.raster_backward(m, g_rgb, g_alpha, g_depth, True, True, True)
    fd, td = _dev(faces), _dev(tex)
    fi, wm, dm = _dev(m["face_index_map"]), _dev(m["weight_map"]), _dev(m["depth_map"])
    gf = torch.zeros_like(fd)
    ops.backward_pixel_map(fd, fi, _dev(m["rgb_map"]), _dev(m["alpha_map"]), _dev(g_rgb), _dev(g_alpha), gf, S, 1e-3, True, True)
    gt = torch.zeros_like(td)
    ops.backward_textures(fi, _dev(m["sampling_weight_map"]), _dev(m["sampling_index_map"]), _dev(g_rgb), gt, F2, faces=fd)
    ops.backward_depth_map(fd, dm, fi, _dev(m["face_inv_map"]), wm, _dev(g_depth), gf, S)
    assert np.isfinite(gf_ref).all() == np.isfinite(gf.cpu().numpy()).all()
    ok = np.isfinite(gf_ref)
    scale = float(np.abs(gf_ref[ok]).max()) if ok.any() else 0.0
    assert np.abs(gf.cpu().numpy()[ok] - gf_ref[ok]).max() <= GRAD_RTOL * scale, (B, S, Fn, size, mode, gmode)
    assert _grad_close(gt.cpu().numpy(), gt_ref)
    # and the forward maps of the same scene, bit for bit, in the form of coverage under test
    fi2 = torch.full((B, S, S), -1, dtype=torch.int32, device="cuda")
    wm2, dm2 = torch.zeros(B, S, S, 3, device="cuda"), torch.full((B, S, S), 3.5, device="cuda")
    with kernels_launched() as k:
        ops.forward_face_index_map(fd, fi2, wm2, dm2, torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda"), S, 0.5, 3.5, 1, 1, 0)
    assert_coverage_form_ran(k.names, coverage)
    assert np.array_equal(fi2.cpu().numpy(), m["face_index_map"]) and np.array_equal(dm2.cpu().numpy(), m["depth_map"])
    assert np.array_equal(wm2.cpu().numpy(), m["weight_map"])


def test_visibility_blobs_do_not_alias():
    """rasterize_ops.visibility() returns a blob of its own per call (round 5: one cached blob per stream, so a second
    forward result on the stream overwrote the first one's list), or fills the caller's `out`."""
    from deep3dmap_amd import _lib
    from deep3dmap_amd.neural_renderer import rasterize_ops as ops
    L = _lib.lib()
    S, F = 32, 40
    fi_a = torch.full((1, S, S), -1, dtype=torch.int32, device="cuda")
    fi_b = fi_a.clone()
    fi_a[0, 3:9, 4:20] = 7
    fi_a[0, 20, 5] = 31
    fi_b[0, 10:12, 10:12] = 2
    blob_a = ops.visibility(fi_a, F)
    keep = blob_a.clone()
    blob_b = ops.visibility(fi_b, F)
    assert blob_a.data_ptr() != blob_b.data_ptr()
    assert torch.equal(blob_a, keep)                          # the second call left the first result alone
    flags_a = blob_a[:4 * F].view(torch.int32).cpu().numpy()
    flags_b = blob_b[:4 * F].view(torch.int32).cpu().numpy()
    assert sorted(np.nonzero(flags_a)[0]) == [7, 31] and sorted(np.nonzero(flags_b)[0]) == [2]
    out = torch.empty(int(L.d3m_visibility_bytes(1, F)), dtype=torch.uint8, device="cuda")
    assert ops.visibility(fi_a, F, out=out) is out
    assert sorted(np.nonzero(out[:4 * F].view(torch.int32).cpu().numpy())[0]) == [7, 31]


def test_scratch_of_a_captured_step_outlives_the_cache():
    """The library's scratch buffers are cached per stream in a BOUNDED cache; a buffer handed out inside a capture is baked
    into the graph and is kept alive by the CapturedStep (rasterize_ops.take_captured_refs), so dropping the cache and
    allocating over it does not touch what the replays write."""
    from deep3dmap_amd import synthetic
    from deep3dmap_amd.graph import CapturedStep
    from deep3dmap_amd.neural_renderer import rasterize_ops as ops
    import deep3dmap_amd.neural_renderer as nr
    v, tri = synthetic.grid_mesh(14)
    v, tri = torch.from_numpy(v)[None].cuda(), torch.from_numpy(tri)[None].cuda()
    vv = v.clone().requires_grad_(True)
    r = nr.Renderer(camera_mode="look_at", image_size=48, anti_aliasing=False)
    r.eye = [0.3, 0.5, -2.5]

    def step():
        vv.grad = None
        sil = r(vv, tri, mode="silhouettes")
        (sil.sum()).backward()
        return sil, vv.grad
    cs = CapturedStep(step).capture()
    assert cs._scratch_refs, "the capture handed out no scratch?"
    sil0, g0 = [t.clone() for t in cs()]
    ops.release_workspaces()
    junk = [torch.full((1 << 22,), 255, dtype=torch.uint8, device="cuda") for _ in range(8)]     # would land in freed scratch
    torch.cuda.synchronize()
    sil1, g1 = [t.clone() for t in cs()]
    assert torch.equal(sil0, sil1) and torch.allclose(g0, g1, rtol=0, atol=1e-6 * float(g0.abs().max()))
    del junk


def test_ops_reject_float64_like_the_reference_extension():
    """rasterize_cuda_kernel.cu:614 dispatches float and double, but every map is read with .data<scalar_t>(): a double
    `faces` with the float32 maps rasterize.py:50-69 allocates raises there.  Same error class here, before any launch."""
    from deep3dmap_amd.neural_renderer import rasterize_ops as ops
    f = torch.zeros(1, 2, 3, 3, device="cuda", dtype=torch.float64)
    fi = torch.zeros(1, 8, 8, device="cuda", dtype=torch.int32)
    m = torch.zeros(1, 8, 8, 3, device="cuda")
    with pytest.raises(RuntimeError, match="Float"):
        ops.forward_face_index_map(f, fi, m, m[..., 0].contiguous(), m, m, 8, 0.1, 100.0, 0, 1, 1)


def test_c_entry_points_reject_bad_arguments_before_any_launch():
    """Error behaviour of the C ABI (include/d3m_raster.h: "0 on success, one of the D3M_ERR_* codes otherwise"), called
    raw: every rejected call returns D3M_ERR_INVALID / D3M_ERR_WORKSPACE and leaves the device alone (a following valid
    call on the same stream succeeds and d3m_last_hip_error() stays 0)."""
    import ctypes
    from deep3dmap_amd import _lib
    L = _lib.lib()
    INVALID, WORKSPACE = 1, 2
    dev = "cuda"
    z = torch.zeros(4096, device=dev)
    zi = torch.zeros(4096, device=dev, dtype=torch.int32)
    p, pi, st = _lib.ptr(z), _lib.ptr(zi), _lib.stream_ptr()
    # the gan2shape block: no struct, a flipped batch of odd size, a raster beyond the candidate walk's 13-bit coordinates,
    # a missing buffer, a view of an unknown layout
    assert L.d3m_g2s_forward(None, st) == INVALID and L.d3m_g2s_backward(None, st) == INVALID
    blk, cam0 = _lib.D3MG2SBlock(), _lib.D3MCamera()          # complete but for the one defect of each case
    blk.batch_size, blk.height, blk.width, blk.image_size = 3, 8, 8, 8
    blk.inv_K_batch = blk.K_batch = 1
    blk.camera = ctypes.pointer(cam0)
    for name in ("inv_K", "K", "rot", "trans", "depth", "albedo", "light_a", "light_b", "light_d", "normal", "diffuse_shading",
                 "texture", "recon_depth", "recon_im", "recon_im_mask", "losses", "screen_vertices", "zbuffer", "scratch"):
        setattr(blk, name, p)
    blk.flip = 1
    assert L.d3m_g2s_forward(ctypes.byref(blk), st) == INVALID                # odd batch with flip3
    blk.flip, blk.image_size = 0, 8193
    assert L.d3m_g2s_forward(ctypes.byref(blk), st) == INVALID                # S > 8192
    blk.image_size, blk.zbuffer = 8, None
    assert L.d3m_g2s_forward(ctypes.byref(blk), st) == INVALID                # a required buffer is missing
    blk.zbuffer, blk.view, blk.view_components = p, p, 4
    assert L.d3m_g2s_forward(ctypes.byref(blk), st) == INVALID                # view of 4 components
    assert L.d3m_g2s_scratch_floats(0, 8, 8, 8) == 0 and L.d3m_g2s_scratch_floats(2, 8, 8, 8) > 0
    # the view-synthesis resampler
    assert L.d3m_warp_resample(None, p, 1, p, 1, p, p, 1.0, p, 3, None, 0, p, None, 1, 8, 8, 8, 8, st) == INVALID
    assert L.d3m_warp_resample(p, p, 2, p, 1, p, p, 1.0, p, 3, None, 0, p, None, 4, 8, 8, 8, 8, st) == INVALID     # inv_K batch 2 of 4
    assert L.d3m_warp_resample(p, p, 1, p, 1, p, p, 1.0, p, 0, None, 0, p, None, 1, 8, 8, 8, 8, st) == INVALID     # no channels
    assert L.d3m_warp_resample_partials(0, 8) <= 0 < L.d3m_warp_resample_partials(8, 8)
    # coverage on an index-free mesh: the row length must divide the vertex count and fix the triangle count
    args = lambda tri, tb, V, Ft: (p, tri, tb, V, Ft, 1, p, pi, p, p, None, 1, 8, 0.1, 100.0, p, z.numel() * 4, None, 0, 0, st)
    assert L.d3m_forward_face_index_map_mesh(*args(None, 1, 16, 18)) == INVALID        # neither indices nor a row length
    assert L.d3m_forward_face_index_map_mesh(*args(None, -5, 16, 18)) == INVALID       # 16 vertices are no rows of 5
    assert L.d3m_forward_face_index_map_mesh(*args(None, -4, 16, 17)) == INVALID       # a 4x4 grid has 18 triangles
    assert L.d3m_forward_face_index_map_mesh(*args(pi, 3, 16, 18)) == INVALID          # index batch 3 of 1
    tiny = (p, None, -4, 16, 18, 1, p, pi, p, p, None, 1, 8, 0.1, 100.0, p, 64, None, 0, 0, st)
    assert L.d3m_forward_face_index_map_mesh(*tiny) == WORKSPACE
    # the mode outputs of its last pass and the reference's face_inv_map exclude each other
    assert L.d3m_forward_face_index_map_mesh_modes(p, None, -4, 16, 18, 1, p, pi, p, p, p, 1, 8, 0.1, 100.0, p, z.numel() * 4,
                                                   None, 0, p, None, None, 0, st) == INVALID
    # the mesh form of the depth backward: needs its vertex target, its visibility blob and its counter
    vt0 = _lib.D3MVertexTarget(p, pi, 16, 18, 1, 1)
    assert L.d3m_backward_depth_map_mesh(p, p, pi, p, p, 1, 36, 8, None, p, p, 0, st) == INVALID
    assert L.d3m_backward_depth_map_mesh(p, p, pi, p, p, 1, 36, 8, ctypes.byref(vt0), None, p, 0, st) == INVALID
    assert L.d3m_backward_depth_map_mesh(p, p, pi, p, p, 1, 36, 8, ctypes.byref(vt0), p, None, 0, st) == INVALID
    assert L.d3m_backward_depth_map_mesh(p, p, pi, p, p, 0, 36, 8, ctypes.byref(vt0), p, p, 0, st) == INVALID
    # the camera's adjoint onto an existing gradient; the objective's scratch for small rasters
    cam = _lib.D3MCamera()
    assert L.d3m_camera_backward_add(p, 3, ctypes.byref(cam), p, p, 2, 4, st) == INVALID    # 3 meshes for 2 views
    assert L.d3m_camera_backward_add(p, 1, None, p, p, 2, 4, st) == INVALID
    for S in (1, 2, 8, 15, 16, 64):
        assert L.d3m_render_fit_scratch_floats(2, S) >= 8 + 4 * 2
    assert L.d3m_last_hip_error() == 0
    # ... and the stream is still usable: a valid call right behind the rejected ones
    sv = torch.rand(1, 16, 3, device=dev) * 1.6 - 0.8
    sv[..., 2] = 1.0 + sv[..., 2].abs()
    faces = torch.empty(1, 36, 3, 3, device=dev)
    fi = torch.empty(1, 8, 8, dtype=torch.int32, device=dev)
    wm, dm = torch.empty(1, 8, 8, 3, device=dev), torch.empty(1, 8, 8, device=dev)
    ws = torch.empty(int(L.d3m_forward_workspace_bytes(1, 36, 8)), dtype=torch.uint8, device=dev)
    rc = L.d3m_forward_face_index_map_mesh(_lib.ptr(sv), None, -4, 16, 18, 1, _lib.ptr(faces), _lib.ptr(fi), _lib.ptr(wm), _lib.ptr(dm),
                                           None, 1, 8, 0.1, 100.0, _lib.ptr(ws), ws.numel(), None, 0, 0, st)
    torch.cuda.synchronize()
    assert rc == 0 and int((fi >= 0).sum()) > 0 and L.d3m_last_hip_error() == 0


@pytest.mark.parametrize("S,V,Ft,size,fill_back", [(64, 300, 500, 0.05, True), (100, 200, 300, 0.3, False), (37, 40, 64, 1.5, True),
                                                    (256, 30, 24, 2.5, True), (129, 2000, 6000, 0.02, True)])
def test_indexed_mesh_coverage_equals_the_operator_on_gathered_faces(S, V, Ft, size, fill_back, coverage):
    """d3m_forward_face_index_map_mesh in one form of coverage (`coverage`: the per-tile lists, or BIDDING -- csrc/d3m_bid.h:
    rows of the boxes, a pass of its own for big faces) against d3m_forward_face_index_map on the gathered faces in the
    OTHER form (each bit-equal to the oracle's brute force and to the reference's kernel above): the same maps, bit for
    bit, on needles, sub-pixel and screen-filling triangles, rasters that are no multiple of the tile, a face with a NaN
    vertex, a zero-area face, and the dense copy / visibility marks both forms leave behind.  Which form ran on either
    side is read from the launch record."""
    import ctypes
    from deep3dmap_amd import _lib
    from deep3dmap_amd.neural_renderer import rasterize_ops as ops
    L = _lib.lib()
    rng = np.random.default_rng(S * 7919 + V)
    B = 2
    verts = np.concatenate([rng.uniform(-1.2, 1.2, (B, V, 2)), rng.uniform(0.3, 4.0, (B, V, 1))], -1).astype(np.float32)
    # triangles of neighbouring-in-space vertices (size controls their extent), plus a few long needles
    centre = rng.uniform(-1.1, 1.1, (Ft, 1, 2))
    d2 = ((verts[0, None, :, :2] - centre) ** 2).sum(-1)                       # [Ft, V]
    near = np.argsort(d2 + rng.uniform(0, size * size, d2.shape), 1)[:, :3]
    tri = near.astype(np.int32)
    tri[::17, 2] = rng.integers(0, V, len(tri[::17]))                            # needles across the image
    tri[5] = (tri[5, 0], tri[5, 0], tri[5, 1])                                   # zero area
    verts[1, tri[9, 0], 0] = np.nan                                              # a NaN vertex in the second view
    sv, tr = _dev(verts), torch.from_numpy(tri[None]).cuda()
    Fp = 2 * Ft if fill_back else Ft
    faces = torch.empty(B, Fp, 3, 3, device="cuda")
    fi = torch.empty(B, S, S, dtype=torch.int32, device="cuda")
    wm, dm, fim = torch.empty(B, S, S, 3, device="cuda"), torch.empty(B, S, S, device="cuda"), torch.empty(B, S, S, 3, 3, device="cuda")
    ws = torch.empty(int(L.d3m_forward_workspace_bytes(B, Fp, S)), dtype=torch.uint8, device="cuda")
    vis = torch.empty(int(L.d3m_visibility_bytes(B, Fp)), dtype=torch.uint8, device="cuda")
    with kernels_launched() as k:
        _lib.check(L.d3m_forward_face_index_map_mesh(_lib.ptr(sv), _lib.ptr(tr), 1, V, Ft, int(fill_back), _lib.ptr(faces), _lib.ptr(fi),
                                                     _lib.ptr(wm), _lib.ptr(dm), _lib.ptr(fim), B, S, 0.5, 3.5, _lib.ptr(ws), ws.numel(),
                                                     _lib.ptr(vis), vis.numel(), 0, _lib.stream_ptr()), "d3m_forward_face_index_map_mesh")
    assert_coverage_form_ran(k.names, coverage)
    _lib.check(L.d3m_visibility(None, _lib.ptr(vis), vis.numel(), B, Fp, S, _lib.stream_ptr()), "d3m_visibility")
    # the reference composition: vertices_to_faces (+ the reversed copies), then the operator
    g = torch.from_numpy(verts)[:, torch.from_numpy(tri).long()]                 # [B,Ft,3,3]
    dense = torch.cat([g, g.flip(2)], 1) if fill_back else g
    fd = dense.contiguous().cuda()
    fi2 = torch.full((B, S, S), -1, dtype=torch.int32, device="cuda")
    wm2, dm2, fim2 = torch.zeros(B, S, S, 3, device="cuda"), torch.full((B, S, S), 3.5, device="cuda"), torch.zeros(B, S, S, 3, 3, device="cuda")
    other = "bidding" if coverage == "binned" else "binned"
    with _lib.coverage_form(other), kernels_launched() as k:
        ops.forward_face_index_map(fd, fi2, wm2, dm2, fim2, torch.zeros_like(fd), S, 0.5, 3.5, 0, 1, 1)
    assert_coverage_form_ran(k.names, other)
    cov = float((fi2 >= 0).float().mean())
    assert cov > 0.02, cov
    assert torch.equal(fi, fi2), int((fi != fi2).sum())
    for a, b in ((wm, wm2), (dm, dm2), (fim, fim2)):
        assert np.array_equal(a.cpu().numpy(), b.cpu().numpy(), equal_nan=True)
    # the dense copy holds every face that owns a pixel, as gathered; the visibility list names exactly those faces
    owners = torch.unique((fi2.long() + torch.arange(B, device="cuda")[:, None, None] * Fp)[fi2 >= 0])
    assert np.array_equal(faces.reshape(-1, 9)[owners].cpu().numpy(), fd.reshape(-1, 9)[owners].cpu().numpy(), equal_nan=True)
    flags = vis.view(torch.int32)[:B * Fp]                                       # the blob starts with one flag per face
    assert torch.equal(flags.nonzero().flatten(), owners)


def test_first_and_last_launch_of_a_lit_step_through_the_c_abi():
    """d3m_lit_front (camera basis + transform, per-face light, clears as ONE launch) against the operators it stands for --
    d3m_camera_basis + d3m_camera_forward + d3m_face_light + d3m_zero_ranges: the same floats bit for bit (shared and per-view
    cameras, shared and per-view meshes, a mesh smaller than a workgroup), ranges that start off a 16-byte boundary cleared
    exactly; d3m_lit_back against d3m_face_light_backward + d3m_camera_backward_add (float atomics: to rounding); bad
    arguments rejected before any launch."""
    import ctypes
    from deep3dmap_amd import _lib
    from deep3dmap_amd.neural_renderer import cameras
    L = _lib.lib()
    dev = torch.device("cuda")
    st = _lib.stream_ptr()
    INVALID = 1
    for B, V, Ft, vb, eyes_b in ((3, 700, 1100, 1, 3), (2, 40, 50, 2, 1), (5, 300, 400, 1, 5)):
        rng = np.random.default_rng(B * 100 + V)
        verts = torch.from_numpy(rng.uniform(-0.8, 0.8, (vb, V, 3)).astype(np.float32)).to(dev)
        tri = torch.from_numpy(rng.integers(0, V, (1, Ft, 3)).astype(np.int32)).to(dev)
        eye = torch.from_numpy((rng.uniform(-1, 1, (eyes_b, 3)) + np.array([0, 0, -2.7])).astype(np.float32)).to(dev)
        p = cameras.look_at_params(verts.expand(B, -1, -1) if vb == 1 and False else verts, eye, _perspective_angle=30)
        p["batch"] = B
        cam, keep = cameras._camera_struct(p, dev)
        want_sv = torch.empty(B, V, 3, device=dev)
        _lib.check(L.d3m_camera_forward(_lib.ptr(verts), vb, ctypes.byref(cam), _lib.ptr(want_sv), B, V, st), "camera")
        Fp = 2 * Ft
        want_light = torch.empty(vb, Fp, 3, device=dev)
        ca = (ctypes.c_float * 3)(1, 0.9, 0.8); cd = (ctypes.c_float * 3)(0.7, 1, 0.6); di = (ctypes.c_float * 3)(0.3, 0.8, -0.5)
        _lib.check(L.d3m_face_light(_lib.ptr(verts), vb, _lib.ptr(tri), 1, _lib.ptr(want_light), 0.4, 0.6, ca, cd, di, vb, V, Ft, 1,
                                    st), "light")
        # the front launch: basis deferred, two ranges to clear (one starting 4 bytes into an allocation)
        pd = cameras.look_at_params(verts, eye, _perspective_angle=30, defer_basis=True)
        pd["batch"] = B
        camd, keepd = cameras._camera_struct(pd, dev)
        basis, bkeep = cameras.basis_struct(pd)
        sv = torch.full((B, V, 3), 7.0, device=dev)
        light = torch.full((vb, Fp, 3), 7.0, device=dev)
        z1, z2 = torch.full((1031,), 5.0, device=dev), torch.full((77,), 5.0, device=dev)
        zp = (ctypes.c_void_p * 2)(z1.data_ptr() + 4, z2.data_ptr())
        zb = (ctypes.c_size_t * 2)(1029 * 4, 77 * 4)
        _lib.check(L.d3m_lit_front(_lib.ptr(verts), vb, ctypes.byref(camd), ctypes.byref(basis), _lib.ptr(sv), B, V, _lib.ptr(tri), 1,
                                   Ft, 1, _lib.ptr(light), vb, 0.4, 0.6, ca, cd, di, zp, zb, 2, st), "d3m_lit_front")
        assert torch.equal(sv, want_sv) and torch.equal(light, want_light), (B, V)
        assert torch.equal(pd["rot"], p["rot"])                                   # the basis left for the adjoint
        assert float(z1[0]) == 5.0 and float(z1[1:1030].abs().max()) == 0.0 and float(z1[1030]) == 5.0 and float(z2.abs().max()) == 0.0
        # the last launch
        g_sv = torch.from_numpy(rng.normal(size=(B, V, 3)).astype(np.float32)).to(dev)
        g_light = torch.from_numpy(rng.normal(size=(vb, Fp, 3)).astype(np.float32)).to(dev)
        want = torch.zeros(vb, V, 3, device=dev)
        _lib.check(L.d3m_face_light_backward(_lib.ptr(verts), vb, _lib.ptr(tri), 1, _lib.ptr(g_light), _lib.ptr(want), 0.4, 0.6, ca, cd,
                                             di, vb, V, Ft, 1, st), "light backward")
        _lib.check(L.d3m_camera_backward_add(_lib.ptr(verts), vb, ctypes.byref(cam), _lib.ptr(g_sv), _lib.ptr(want), B, V, st), "camera backward")
        got = torch.zeros(vb, V, 3, device=dev)
        _lib.check(L.d3m_lit_back(_lib.ptr(verts), vb, ctypes.byref(cam), _lib.ptr(g_sv), _lib.ptr(got), B, V, _lib.ptr(tri), 1, Ft, 1,
                                  _lib.ptr(g_light), vb, 0.4, 0.6, ca, cd, di, st), "d3m_lit_back")
        assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max()), (B, V)
    # rejected before any launch
    a = torch.zeros(64, device=dev)
    pa = _lib.ptr(a)
    assert L.d3m_lit_front(None, 1, None, None, None, 1, 4, None, 1, 0, 0, None, 1, 0.5, 0.5, None, None, None, None, None, 0, st) == INVALID
    assert L.d3m_lit_front(pa, 1, None, None, None, 1, 4, None, 1, 0, 0, None, 1, 0.5, 0.5, None, None, None, None, None, 11, st) == INVALID   # too many ranges
    cam0 = _lib.D3MCamera()
    assert L.d3m_lit_front(pa, 1, ctypes.byref(cam0), None, None, 1, 4, None, 1, 0, 0, None, 1, 0.5, 0.5, None, None, None, None, None, 0, st) == INVALID  # no output
    assert L.d3m_lit_back(pa, 1, None, pa, pa, 1, 4, pa, 1, 2, 0, pa, 1, 0.5, 0.5, pa, pa, pa, st) == INVALID                  # no camera
    zb1 = (ctypes.c_size_t * 1)(6)
    zp1 = (ctypes.c_void_p * 1)(a.data_ptr())
    assert L.d3m_lit_front(pa, 1, None, None, None, 1, 4, None, 1, 0, 0, None, 1, 0.5, 0.5, None, None, None, zp1, zb1, 1, st) == INVALID   # not a multiple of 4 bytes
    assert L.d3m_last_hip_error() == 0


def test_pooled_fit_records_through_the_c_abi():
    """d3m_render_lit_epilogue with anti-aliasing and a fused objective, called raw, in its two gradient forms on the same
    coverage: unscaled gradient MAPS (round 4's form: fit->grad_*_map) and per-pixel RECORDS (D3M_FIT_POOLED, round 5).
    Same pooled images bit for bit, same objective (to the order of its partial sums), and the records are the maps with the
    scalars the readers no longer apply -- 1 / S^2 on alpha, 1 / (3 mask_sum) on rgb -- plus <values, gradient>, the owner
    and the lines' non-zero extents.  Without the flag the records form of an anti-aliased pass is refused."""
    import ctypes
    from deep3dmap_amd import _lib
    L = _lib.lib()
    dev, st = "cuda", _lib.stream_ptr()
    gen = torch.Generator().manual_seed(5)
    row, S, ts = 12, 64, 2
    V, Ft = row * row, 2 * (row - 1) * (row - 1)
    Fp, s = 2 * Ft, S // 2
    yy, xx = torch.meshgrid(torch.linspace(-0.8, 0.8, row), torch.linspace(-0.8, 0.8, row), indexing="ij")
    sv = torch.stack([xx + 0.03 * torch.randn(row, row, generator=gen), yy + 0.03 * torch.randn(row, row, generator=gen),
                      1.5 + 0.2 * torch.rand(row, row, generator=gen)], -1).reshape(1, V, 3).contiguous().to(dev)
    faces = torch.empty(1, Fp, 3, 3, device=dev)
    fi = torch.empty(1, S, S, dtype=torch.int32, device=dev)
    wm, dm = torch.empty(1, S, S, 3, device=dev), torch.empty(1, S, S, device=dev)
    ws = torch.empty(int(L.d3m_forward_workspace_bytes(1, Fp, S)), dtype=torch.uint8, device=dev)
    _lib.check(L.d3m_forward_face_index_map_mesh(_lib.ptr(sv), None, -row, V, Ft, 1, _lib.ptr(faces), _lib.ptr(fi), _lib.ptr(wm),
                                                 _lib.ptr(dm), None, 1, S, 0.1, 100.0, _lib.ptr(ws), ws.numel(), None, 0, 0, st),
               "d3m_forward_face_index_map_mesh")
    assert 0.3 < float((fi >= 0).float().mean()) < 0.95          # a silhouette inside the image
    tex = torch.rand(1, Ft, ts, ts, ts, 3, generator=gen).to(dev)
    light = (0.3 + 0.7 * torch.rand(1, Fp, 3, generator=gen)).to(dev)
    bg = torch.tensor([[0.1, 0.2, 0.3]], device=dev)
    rgb_t = torch.rand(1, 3, s, s, generator=gen).to(dev)
    depth_t = (1.4 + 0.3 * torch.rand(1, s, s, generator=gen)).to(dev)
    alpha_t = (torch.rand(1, s, s, generator=gen) > 0.4).float().to(dev)
    mask = (torch.rand(1, s, s, generator=gen) > 0.2).float().to(dev)
    mask_sum = mask.sum().reshape(1)

    def run(records, flags):
        out = {"rgb_blended": torch.empty(1, S, S, 3, device=dev), "alpha_map": torch.empty(1, S, S, device=dev),
               "rgb": torch.empty(1, 3, s, s, device=dev), "alpha": torch.empty(1, s, s, device=dev),
               "depth": torch.empty(1, s, s, device=dev), "loss": torch.zeros(1, device=dev),
               "g_depth": torch.empty(1, S, S, device=dev),
               "scratch": torch.empty(int(L.d3m_render_fit_scratch_floats(1, S)), device=dev)}
        if records:
            out.update(grad=torch.empty(1, S, S, 4, device=dev), dot=torch.empty(1, S, S, 2, device=dev),
                       nz=torch.zeros(2, 1, 2, S, dtype=torch.int32, device=dev))
            maps = (None, None, out["g_depth"])
            rec = (out["grad"], out["dot"], out["nz"][0], out["nz"][1])
        else:
            out.update(g_rgb=torch.empty(1, S, S, 3, device=dev), g_alpha=torch.empty(1, S, S, device=dev))
            maps = (out["g_rgb"], out["g_alpha"], out["g_depth"])
            rec = (None, None, None, None)
        fit = _lib.D3MFitTargets(_lib.ptr(rgb_t), _lib.ptr(depth_t), _lib.ptr(alpha_t), _lib.ptr(mask), _lib.ptr(out["scratch"]),
                                 _lib.ptr(out["loss"]), *[_lib.ptr(t) for t in maps], None, _lib.ptr(mask_sum),
                                 *[_lib.ptr(t) for t in rec], flags)
        rc = L.d3m_render_lit_epilogue(_lib.ptr(faces), _lib.ptr(tex), 1, _lib.ptr(light), 1, _lib.ptr(fi), _lib.ptr(wm), _lib.ptr(dm),
                                       _lib.ptr(bg), 1, _lib.ptr(out["rgb_blended"]), _lib.ptr(out["alpha_map"]), _lib.ptr(out["rgb"]),
                                       _lib.ptr(out["alpha"]), _lib.ptr(out["depth"]), 1, Ft, 1, S, ts, 1e-5, 1,
                                       ctypes.byref(fit), st)
        torch.cuda.synchronize()
        return rc, out

    assert run(True, 0)[0] == 1                                   # D3M_ERR_INVALID: records of a pooled pass without the flag
    rc_m, a = run(False, 0)
    rc_r, b = run(True, _lib.FIT_POOLED)
    assert rc_m == 0 and rc_r == 0 and L.d3m_last_hip_error() == 0
    for name in ("rgb", "alpha", "depth", "rgb_blended", "alpha_map", "g_depth"):
        assert torch.equal(a[name], b[name]), name
    assert float(a["loss"]) > 0 and abs(float(a["loss"]) - float(b["loss"])) <= 2e-6 * float(a["loss"])
    # the records against the maps
    grad, dot = b["grad"][0], b["dot"][0]
    assert torch.allclose(grad[..., 0] * float(S * S), a["g_alpha"][0], rtol=1e-6, atol=0)
    assert torch.allclose(grad[..., 1:] * (3.0 * float(mask_sum)), a["g_rgb"][0], rtol=1e-6, atol=1e-12)
    want = b["alpha_map"][0] * grad[..., 0] + (b["rgb_blended"][0] * grad[..., 1:]).sum(-1)
    assert torch.allclose(dot[..., 0], want, rtol=1e-5, atol=1e-9)
    assert torch.equal(dot[..., 1].contiguous().view(torch.int32), fi[0])
    nzp = ((grad != 0).any(-1) | (dot[..., 0] != 0)).cpu().numpy()              # [y, x], row 0 = bottom
    assert nzp.any() and not nzp.all()
    lo_inv, hi1 = b["nz"][0][0].cpu().numpy(), b["nz"][1][0].cpu().numpy()       # [axis, line]
    for axis, lines in ((1, nzp), (0, nzp.T)):                                    # axis 1: rows (along x), axis 0: columns
        for d0 in range(S):
            idx = np.flatnonzero(lines[d0])
            exp = (0, 0) if idx.size == 0 else (S - int(idx[0]), int(idx[-1]) + 1)
            assert (int(lo_inv[axis, d0]), int(hi1[axis, d0])) == exp, (axis, d0)
