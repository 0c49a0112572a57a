"""GPU parity tests against THE REFERENCE'S OWN KERNELS run on the device (oracle/_ref/libnr_ref_hip*.so: the text of
rasterize_cuda_kernel.cu:22-593 compiled by hipcc where it lies, see oracle/Makefile `ref_hip`), at fixture size and at
the full size of BASELINE.json's configurations 2, 4 and 5.

Tolerances (north_star): forward maps are held to EQUALITY against the -ffp-contract=off build of the reference
(indices exact, floats bit for bit) and to 1e-4 on depth / barycentrics against its default (FMA-contracted) build away
from the handful of pixels whose winner flips with the contraction; backward gradients to 1e-3 of each tensor's scale.
"""
import ctypes

import numpy as np
import pytest
import torch

from conftest import assert_coverage_form_ran, kernels_launched
from oracle import nr_oracle as O
from oracle import nr_ref_hip as RH

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not RH.available(), reason="oracle/_ref/libnr_ref_hip.so not built (make -C oracle ref_hip)")]

GRAD_RTOL = 1e-3        # north_star: 1e-3 on backward gradients, relative to the tensor's largest entry
FWD_TOL = 1e-4          # north_star: 1e-4 on depth / barycentrics (used only against the FMA build)


def _rel_max(got, ref):
    ref = ref.float()
    return float((got.float() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


ELEM_RTOL, ELEM_ATOL_OF_MAX = 1e-3, 1e-5     # element-wise: |d| <= 1e-3 |ref| + 1e-5 max|ref|
ELEM_VIOLATING_FRAC = 7e-6                  # ... for all but this fraction of the entries (measured: 0.3 .. 2.1e-6, x 3)


def _grad_stats(got, ref):
    """How a gradient tensor's error is DISTRIBUTED (a bound relative to the largest entry alone lets
    an entry 100x below the maximum be off by 4 %): the largest deviation over the largest entry, the fraction of entries
    violating |d| <= 1e-3 |ref| + 1e-5 max|ref|, the 99.9th percentile and the maximum of |d| / (|ref| + 1e-5 max|ref|)
    over the entries the reference writes, and where the worst entry sits."""
    got, ref = got.float().reshape(-1), ref.float().reshape(-1)
    scale = float(ref.abs().max().clamp_min(1e-30))
    d = (got - ref).abs()
    viol = d > ELEM_RTOL * ref.abs() + ELEM_ATOL_OF_MAX * scale
    nz = ref != 0
    rel = (d[nz] / (ref[nz].abs() + ELEM_ATOL_OF_MAX * scale)).sort().values
    worst = int(d.argmax())
    return {"rel_max": float(d.max() / scale), "violating_frac": float(viol.float().mean()),
            "violating_frac_of_written": float(viol[nz].float().mean()) if int(nz.sum()) else 0.0,
            "rel_p999": float(rel[min(rel.numel() - 1, int(0.999 * rel.numel()))]) if rel.numel() else 0.0,
            "rel_elementwise_max": float(rel[-1]) if rel.numel() else 0.0, "entries": int(ref.numel()),
            "entries_written": int(nz.sum()), "worst_index": worst,
            "worst_ref_over_max": float(ref[worst].abs() / scale), "worst_rel_to_itself": float(d[worst] / ref[worst].abs().clamp_min(1e-30))}


# ------------------------------------------------------------------------------------------------------------
# 1. the CPU oracle (port) is the reference: fresh random scenes, port vs device reference, every kernel
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("trial", range(8))
def test_cpu_oracle_equals_device_reference_on_random_scenes(trial):
    rng = np.random.default_rng(990 + trial)
    B, Fn = int(rng.integers(1, 3)), int(rng.integers(4, 64))
    S, ts = int(rng.choice([16, 24, 40, 57])), int(rng.choice([1, 2, 3]))
    size = float(rng.choice([0.1, 0.7, 2.0]))
    xy = rng.uniform(-1.3, 1.3, (B, Fn, 1, 2)) + rng.uniform(-size, size, (B, Fn, 3, 2))
    faces = np.concatenate([xy, rng.uniform(0.3, 4.0, (B, Fn, 3, 1))], -1).astype(np.float32)
    if ts == 1:
        faces[:, -3:, :, 0] += 10.0        # KCU:229-233 reads the next faces' texels for ts=1; keep that inside the buffer
    if trial % 2:
        faces = np.concatenate([faces, faces[:, :, ::-1]], 1).copy()
    tex = rng.uniform(0, 1, faces.shape[:2] + (ts, ts, ts, 3)).astype(np.float32)
    grads = [rng.normal(size=s).astype(np.float32) for s in ((B, S, S, 3), (B, S, S), (B, S, S))]
    res = {}
    for be in ("port", "hip"):
        m = O.raster_forward(faces, tex, S, 0.5, 3.5, 1e-3, (0.2, 0.1, 0.3), True, True, True, backend=be)
        gf4, _ = O.raster_backward(m, grads[0], grads[1], None, True, True, False, backend=be)
        gf, gt = O.raster_backward(m, *grads, True, True, True, backend=be)
        res[be] = (m, gf4, gf, gt)
    for key in ("face_index_map", "weight_map", "depth_map", "rgb_map", "alpha_map", "face_inv_map", "faces_inv",
                "sampling_index_map", "sampling_weight_map"):
        assert np.array_equal(res["port"][0][key], res["hip"][0][key], equal_nan=True), (trial, key)
    # K4 is one thread per face, no atomics: same operations, same order
    a, b = res["port"][1], res["hip"][1]
    assert np.array_equal(np.isfinite(a), np.isfinite(b))
    ok = np.isfinite(a)
    assert np.abs(a[ok] - b[ok]).max() <= 1e-5 * max(1e-30, np.abs(b[ok]).max()), (trial, "K4")
    # K5 / K6 sum with float atomics on the device: equal up to the order of the additions
    for i, name in ((2, "grad_faces"), (3, "grad_textures")):
        a, b = res["port"][i], res["hip"][i]
        ok = np.isfinite(a) & np.isfinite(b)
        assert np.abs(a[ok] - b[ok]).max() <= 1e-5 * max(1e-30, np.abs(b[ok]).max()), (trial, name)


# ------------------------------------------------------------------------------------------------------------
# 2. the product's five operators against the device reference: the fuzz family of test_gpu_ops.py
# ------------------------------------------------------------------------------------------------------------
def _product_forward(faces, tex, S, near, far, eps, background):
    from deep3dmap_amd.neural_renderer.rasterize import _background_tensor, _epilogue, _raster_forward
    bg = _background_tensor(background, faces.device)
    m, rgb_sampled = _raster_forward(faces, tex, S, near, far, eps, bg, True, True, True, True)
    _epilogue(m, rgb_sampled, bg, faces.shape[0], S, False, True, True, True, False)
    return m


def _product_backward(faces, tex, m, S, eps, g_rgb, g_alpha, g_depth):
    from deep3dmap_amd.neural_renderer.rasterize import _raster_backward
    return _raster_backward(faces, tex, m, S, eps, g_rgb, g_alpha, g_depth, True, True, True, True)


def _raster_backward_parts(faces, tex, m, S, eps, g_rgb, g_alpha, g_depth):
    """the product's backward with only some of the output gradients (K4 alone / K6 alone)"""
    from deep3dmap_amd.neural_renderer.rasterize import _raster_backward
    rr, rd = g_rgb is not None, g_depth is not None
    return _raster_backward(faces, tex, m, S, eps, g_rgb, g_alpha, g_depth, rr, rr, rd, False)


def _assert_maps_equal(m, ref, keys=("face_index_map", "weight_map", "depth_map", "rgb_map", "alpha_map")):
    for k in keys:
        a, b = m[k], ref[k]
        if a.dtype.is_floating_point:       # bit for bit, NaN == NaN
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)) or \
                torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0)), \
                (k, int((a != b).sum()), float((a - b).abs().max()))
        else:
            assert torch.equal(a, b), (k, int((a != b).sum()))


@pytest.mark.parametrize("seed", range(40))
def test_product_operators_against_device_reference_fuzz(seed, coverage):
    """Sub-pixel to screen-filling triangles, vertices snapped to pixel centres / edges, duplicated and degenerate
    faces, off-screen faces, dense / boxed / single-pixel gradient maps -- product vs the reference's kernels, once per
    form of coverage (`coverage`: per-tile lists | bidding)."""
    rng = np.random.default_rng(5000 + seed)
    B = int(rng.integers(1, 3))
    S = int(rng.choice([16, 24, 33, 48, 64, 100]))
    Fn = int(rng.integers(1, 60))
    size = float(rng.choice([0.03, 0.15, 0.6, 2.0]))
    xy = rng.uniform(-1.2, 1.2, (B, Fn, 1, 2)) + rng.uniform(-size, size, (B, Fn, 3, 2))
    mode = seed % 4
    if mode == 1:
        xy = (2 * np.round((xy * S + S - 1) / 2) + 1 - S) / S
    elif mode == 2:
        xy = np.round(xy * S / 2) * 2 / S
    faces = np.concatenate([xy, rng.uniform(0.3, 4.0, (B, Fn, 3, 1))], -1).astype(np.float32)
    if Fn > 3:
        faces[:, -1] = faces[:, 0]
        faces[:, -2, 1] = faces[:, -2, 0]
    faces = np.concatenate([faces, faces[:, :, ::-1]], 1).copy()
    ts = int(rng.choice([2, 2, 3]))
    fd = torch.from_numpy(faces).cuda()
    td = torch.rand(B, faces.shape[1], ts, ts, ts, 3, device="cuda")
    ref = RH.forward(fd, td, S, 0.5, 3.5, 1e-3, (0.1, 0.2, 0.3))
    with kernels_launched() as k:
        m = _product_forward(fd, td, S, 0.5, 3.5, 1e-3, (0.1, 0.2, 0.3))
    assert_coverage_form_ran(k.names, coverage)
    _assert_maps_equal(m, ref, ("face_index_map", "weight_map", "depth_map", "rgb_map", "alpha_map", "face_inv_map",
                                "sampling_index_map", "sampling_weight_map"))
    assert np.array_equal(m["faces_inv"].cpu().numpy(), ref["faces_inv"].cpu().numpy(), equal_nan=True)     # K1, KCU:24-67
    g_rgb = torch.randn(B, S, S, 3, device="cuda")
    g_alpha = torch.randn(B, S, S, device="cuda")
    g_depth = torch.randn(B, S, S, device="cuda")
    gmode = (seed // 4) % 3
    if gmode == 1:
        box = torch.zeros(B, S, S, device="cuda")
        box[:, S // 4: 3 * S // 4, S // 3: S - 2] = 1
        g_rgb, g_alpha = g_rgb * box[..., None], g_alpha * box
    elif gmode == 2:
        one = torch.zeros(B, S, S, device="cuda")
        one[:, S // 2, S // 2] = 1
        g_rgb, g_alpha = g_rgb * 0, g_alpha * one
    gf_ref, gt_ref = RH.backward(ref, g_rgb, g_alpha, g_depth, True, True, True)
    gf, gt = _product_backward(fd, td, m, S, 1e-3, g_rgb, g_alpha, g_depth)
    ok = torch.isfinite(gf_ref)
    assert torch.equal(ok, torch.isfinite(gf))
    if ok.any() and float(gf_ref[ok].abs().max()) > 0:
        assert float((gf[ok] - gf_ref[ok]).abs().max()) <= GRAD_RTOL * float(gf_ref[ok].abs().max()), (B, S, Fn, size, mode, gmode)
    if float(gt_ref.abs().max()) > 0:
        assert _rel_max(gt, gt_ref) <= GRAD_RTOL


@pytest.mark.parametrize("seed", range(24))
def test_sliver_faces_against_device_reference(seed, coverage):
    """The forward's one theoretical deviation (either form bounds a face's pixels by its dilated box): a (near-)zero-area face selected by brute force at
    a pixel outside its dilated bounding box.  800 slivers per scene -- collinear up to rounding, thin, and exactly
    collinear through pixel centres -- product vs the reference's kernels: counted, and the count must be zero."""
    rng = np.random.default_rng(seed)
    S, Fn = int(rng.choice([32, 64, 128])), 400
    p0 = rng.uniform(-1.1, 1.1, (1, Fn, 1, 2))
    d = rng.uniform(-1, 1, (1, Fn, 1, 2)) * float(rng.choice([0.05, 0.3, 1.5]))
    xy = p0 + d * rng.uniform(0, 1, (1, Fn, 3, 1))
    if seed % 3 == 0:
        xy = xy + rng.normal(size=xy.shape) * 1e-7
    elif seed % 3 == 1:
        xy = xy + rng.normal(size=xy.shape) * 1e-4
    else:
        c = (2 * rng.integers(0, S, (1, Fn, 1, 2)) + 1 - S) / S
        xy = c + rng.integers(-3, 4, (1, Fn, 1, 2)) * 2.0 / S * rng.integers(0, 6, (1, Fn, 3, 1))
    faces = np.concatenate([xy, rng.uniform(0.6, 3.0, (1, Fn, 3, 1))], -1).astype(np.float32)
    fd = torch.from_numpy(np.concatenate([faces, faces[:, :, ::-1]], 1).copy()).cuda()
    td = torch.rand(1, fd.shape[1], 2, 2, 2, 3, device="cuda")
    ref = RH.forward(fd, td, S, 0.5, 3.5, 1e-3, (0, 0, 0))
    with kernels_launched() as k:
        m = _product_forward(fd, td, S, 0.5, 3.5, 1e-3, (0, 0, 0))
    assert_coverage_form_ran(k.names, coverage)
    assert int((m["face_index_map"] != ref["face_index_map"]).sum()) == 0
    _assert_maps_equal(m, ref, ("weight_map", "depth_map", "rgb_map"))


# ------------------------------------------------------------------------------------------------------------
# 3. BASELINE.json's configurations at FULL SIZE
# ------------------------------------------------------------------------------------------------------------
def _config_scene(n, eyes, image_size, anti_aliasing):
    """grid_mesh(n) seen from `eyes` through the product's camera + gather (look_at, 30 degrees, fill_back)."""
    from deep3dmap_amd import neural_renderer as nr, synthetic
    from deep3dmap_amd.neural_renderer.mesh_ops import gather_faces
    v, tri = synthetic.grid_mesh(n)
    B = len(eyes)
    eyes = torch.from_numpy(np.ascontiguousarray(eyes)).cuda()
    vt = torch.from_numpy(v).cuda()[None].expand(B, -1, -1).contiguous()
    ft = torch.from_numpy(tri).cuda()[None].expand(B, -1, -1).contiguous()
    sv = nr.look_at(vt, eyes, _perspective_angle=30)
    faces = gather_faces(sv, ft, True)
    S = image_size * 2 if anti_aliasing else image_size
    return v, tri, sv, faces, S


def _record(name, info):
    """Append one configuration's numbers to gpurun_out/parity_full_size.json (copied to profiles/ by hand)."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_full_size.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[name] = info
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)


def _full_size_check(faces, S, ts, form, ran, seed, textures_batch=None):
    """Forward maps EQUAL to the reference's, K4/K5/K6 gradients within 1e-3, on one full-size scene.
    form: the coverage form the caller selected ("binned" | "bidding" | "auto"); ran: the form that must have RUN, checked
    by the names of the launched kernels, not by arithmetic on the tile count."""
    from deep3dmap_amd import _lib
    B, Fp = faces.shape[:2]
    gen = torch.Generator(device="cuda").manual_seed(seed)
    tex = torch.rand(B, Fp, ts, ts, ts, 3, device="cuda", generator=gen)
    near, far, eps, bg = 0.1, 100.0, 1e-3, (0.3, 0.2, 0.1)
    ref = RH.forward(faces, tex, S, near, far, eps, bg)
    with _lib.coverage_form(form), kernels_launched() as k:
        m = _product_forward(faces, tex, S, near, far, eps, bg)
    assert_coverage_form_ran(k.names, ran)
    cov = float((ref["face_index_map"] >= 0).float().mean())
    assert 0.2 < cov < 0.95, cov
    mismatched = int((m["face_index_map"] != ref["face_index_map"]).sum())
    assert mismatched == 0, f"{mismatched} of {B * S * S} pixels pick another face than the reference"
    _assert_maps_equal(m, ref)
    g_rgb = torch.randn(B, S, S, 3, device="cuda", generator=gen)
    g_alpha = torch.randn(B, S, S, device="cuda", generator=gen)
    g_depth = torch.randn(B, S, S, device="cuda", generator=gen)
    gf_ref, gt_ref = RH.backward(ref, g_rgb, g_alpha, g_depth, True, True, True)
    gf, gt = _product_backward(faces, tex, m, S, eps, g_rgb, g_alpha, g_depth)
    assert torch.isfinite(gf_ref).all() and torch.isfinite(gf).all()
    errs = {"grad_faces": _rel_max(gf, gf_ref), "grad_textures": _rel_max(gt, gt_ref)}
    assert errs["grad_faces"] <= GRAD_RTOL and errs["grad_textures"] <= GRAD_RTOL, errs
    # ... and element by element (recorded per configuration and form in the parity file)
    dist = {"grad_faces": _grad_stats(gf, gf_ref), "grad_textures": _grad_stats(gt, gt_ref)}
    # which operator the face gradient's error belongs to: K4 alone (rgb + alpha gradients; x, y entries) and K6 alone
    # (depth gradient); and what kind of face carries the worst entry (its area in pixels: K4 divides by edge lengths,
    # K6 by the inverse's determinant)
    gf4_ref, _ = RH.backward(ref, g_rgb, g_alpha, None, True, True, False)
    gf4, _ = _raster_backward_parts(faces, tex, m, S, eps, g_rgb, g_alpha, None)
    dist["grad_faces_K4_alone"] = _grad_stats(gf4, gf4_ref)
    # HOW MUCH OF THAT IS THE F32 WALK'S OWN ROUNDING.  The reference's kernel instantiated for double (its
    # own second dispatch, KCU:614) on the same inputs is the value both f32 evaluations approximate.  Measured against
    # it, the reference's float instantiation and the product are the same distance away (recorded per configuration): the
    # ~9e-4 tail of product-vs-reference is two independent f32 roundings of sums whose terms cancel, not a drift of the
    # product -- and the product must stay as close to the f64 value as the reference's own f32 evaluation is.
    truth = RH.backward_pixel_map_f64(ref, g_rgb, g_alpha)
    dist["K4_reference_f32_vs_reference_f64"] = _grad_stats(gf4_ref, truth)
    dist["K4_product_vs_reference_f64"] = _grad_stats(gf4, truth)
    noise, ours = dist["K4_reference_f32_vs_reference_f64"], dist["K4_product_vs_reference_f64"]
    assert ours["rel_p999"] <= 1.5 * noise["rel_p999"] + 1e-4 and ours["rel_max"] <= 1.5 * noise["rel_max"] + 1e-4, (ours, noise)
    del truth
    gf6_ref, _ = RH.backward(ref, None, None, g_depth, False, False, True)
    gf6, _ = _raster_backward_parts(faces, tex, m, S, eps, None, None, g_depth)
    dist["grad_faces_K6_alone"] = _grad_stats(gf6, gf6_ref)
    f_worst = dist["grad_faces"]["worst_index"] // 9
    fw = faces.reshape(-1, 3, 3)[f_worst].double()
    area_px = float(((fw[1, 0] - fw[0, 0]) * (fw[2, 1] - fw[0, 1]) - (fw[2, 0] - fw[0, 0]) * (fw[1, 1] - fw[0, 1])).abs() * S * S / 8)
    dist["worst_face"] = {"face": int(f_worst % Fp), "view": int(f_worst // Fp), "area_px": area_px,
                          "pixels_owned": int((ref["face_index_map"].reshape(B, -1)[f_worst // Fp] == f_worst % Fp).sum()),
                          "entry": int(dist["grad_faces"]["worst_index"] % 9),
                          "K4_err_there": float((gf4 - gf4_ref).reshape(-1)[dist["grad_faces"]["worst_index"]].abs()),
                          "K6_err_there": float((gf6 - gf6_ref).reshape(-1)[dist["grad_faces"]["worst_index"]].abs())}
    for name in ("grad_faces", "grad_textures"):
        assert dist[name]["violating_frac"] <= ELEM_VIOLATING_FRAC, (name, dist[name])
    del gf, gt, gf_ref, gt_ref, g_rgb, g_alpha, g_depth, gf4, gf4_ref, gf6, gf6_ref
    # INFORMATIONAL: the default (FMA-contracted) build of the same reference text -- what a stock build of its setup.py
    # computes.  The reference's arithmetic is ill-conditioned on small / sliver triangles (face_inv divides by twice
    # the signed area, KCU:52-61; a 1-pixel triangle's barycentrics amplify one ulp of its vertices by ~1e3), so its own
    # two builds disagree with EACH OTHER: winners flip on edge pixels and barycentrics move by 1e-4..1e-1.  The numbers
    # are recorded (gpurun_out/parity_full_size.json -> profiles/) and only sanity-bounded here; the contract that is
    # asserted is equality with the uncontracted build above, which is stricter than north_star's 1e-4.
    m = {k_: m[k_] for k_ in ("face_index_map", "depth_map", "weight_map")}
    del ref
    fma = RH.forward(faces, tex, S, near, far, eps, bg, contract="fma")
    same = (fma["face_index_map"] == m["face_index_map"])
    covered = same & (m["face_index_map"] >= 0)
    flipped = float((~same).float().mean())
    d_depth = (fma["depth_map"] - m["depth_map"])[covered].abs() / m["depth_map"][covered].abs()
    d_w = (fma["weight_map"] - m["weight_map"])[covered].abs().amax(-1)
    stats = {"fma_flipped_frac": flipped, "fma_depth_over_1e-4_frac": float((d_depth > FWD_TOL).float().mean()),
             "fma_weight_over_1e-4_frac": float((d_w > FWD_TOL).float().mean()),
             "fma_depth_median": float(d_depth.median()), "fma_weight_median": float(d_w.median()),
             "fma_weight_max": float(d_w.max())}
    assert flipped < 1e-3 and stats["fma_depth_median"] < 1e-5 and stats["fma_depth_over_1e-4_frac"] < 5e-3, stats
    return {"coverage": cov, "pixels": B * S * S, "faces": Fp, "mismatched_pixels": mismatched, "coverage_form": ran,
            "coverage_kernels": sorted(n for n in k.names if n.startswith(("k_bin", "k_bid", "k_raster"))), **stats, **errs,
            "gradient_error_distribution": dist}


def test_config4_nine_views_against_reference(coverage):
    """BASELINE config 4's mesh: 100,352 triangles (fill_back: 200,704), 512x512, NINE of the 32 cameras in one launch
    (36,864 tiles), in each form of coverage."""
    from deep3dmap_amd import synthetic
    eyes = synthetic.camera_ring(32)[[0, 3, 7, 11, 14, 18, 22, 26, 29]]
    _, _, _, faces, S = _config_scene(225, eyes, 512, False)
    info = _full_size_check(faces, S, 2, coverage, coverage, 4)
    _record(f"config4_100352tri_512_9views[{coverage}]", info)


def test_config4_bench_launch_shape_auto_dispatch_against_reference():
    """The launch bench.py times: ALL 32 cameras of config 4 in one launch = 131,072 tiles, form of coverage left to the
    library ("auto") -- which must pick the per-tile lists with ONE wave per tile (k_bin_count -> k_bin_alloc -> k_bin_fill
    -> k_raster_tiles<..., 1>: more than 65,536 tiles) -- against the reference's kernels (32 x 5e10 pixel-face tests)."""
    from deep3dmap_amd import _lib, synthetic
    eyes = synthetic.camera_ring(32)
    _, _, _, faces, S = _config_scene(225, eyes, 512, False)
    assert faces.shape[0] * ((S + 7) // 8) ** 2 == 131072      # (blocks of 8 x 8 pixels: twice the library's auto-dispatch threshold)
    assert _lib.lib().d3m_get_coverage_form() == -1
    info = _full_size_check(faces, S, 2, "auto", "binned", 41)
    _record("config4_100352tri_512_32views[auto=binned]", info)


def test_config4_full_size_against_the_cpu_port():
    """Two cameras of config 4 against the CPU oracle (bounding-box forward, 0.2 s per view): maps equal."""
    from deep3dmap_amd import synthetic
    eyes = synthetic.camera_ring(32)[[5, 21]]
    _, _, _, faces, S = _config_scene(225, eyes, 512, False)
    tex = torch.rand(2, faces.shape[1], 2, 2, 2, 3, device="cuda")
    m = _product_forward(faces, tex, S, 0.1, 100.0, 1e-3, (0, 0, 0))
    ref = O.raster_forward(faces.cpu().numpy(), tex.cpu().numpy(), S, 0.1, 100.0, 1e-3, (0, 0, 0), True, True, True,
                           backend="port", bbox=True)
    for k in ("face_index_map", "weight_map", "depth_map", "rgb_map", "alpha_map"):
        assert np.array_equal(m[k].cpu().numpy(), ref[k]), k


def test_config2_full_size_against_reference(coverage):
    """BASELINE config 2: ~53k-triangle mesh (grid_mesh(164): 53,138; fill_back 106,276) @256x256 with anti-aliasing
    (internal raster 512x512), single view (4,096 tiles: the per-tile lists run four waves per tile)."""
    from deep3dmap_amd import synthetic
    eyes = synthetic.camera_ring(8)[[1]]
    _, _, _, faces, S = _config_scene(164, eyes, 256, True)
    assert S == 512 and faces.shape[1] == 106276
    info = _full_size_check(faces, S, 2, coverage, coverage, 2)
    _record(f"config2_53138tri_256aa_1view[{coverage}]", info)


def test_config5_full_size_against_reference(coverage):
    """BASELINE config 5: 1,002,528 triangles (fill_back 2,005,056) @1024x1024, one of the 256 cameras (the reference's
    brute force is 2.1e12 pixel-face tests per view)."""
    from deep3dmap_amd import synthetic
    eyes = synthetic.camera_ring(256)[[37]]
    _, _, _, faces, S = _config_scene(709, eyes, 1024, False)
    assert faces.shape[1] == 2005056
    info = _full_size_check(faces, S, 2, coverage, coverage, 5)
    _record(f"config5_1002528tri_1024_1view[{coverage}]", info)


@pytest.mark.parametrize("views,form,ran", [(9, "binned", "binned"), (9, "bidding", "bidding"), (20, "auto", "binned")])
def test_config4_mesh_path_and_lit_step_against_reference(views, form, ran):
    """The path bench.py actually runs -- coverage straight from the indexed mesh (d3m_forward_face_index_map_mesh),
    fill_back and lighting on the fly, fused backward into vertex gradients -- on nine full-size views in each form of
    coverage, and on twenty (81,920 tiles) with the form left to the library, which must take the per-tile lists: its maps
    against the reference's kernels fed with the materialised faces / lit textures (NR/renderer.py:155-167), and
    d(loss)/d(screen vertices), d(loss)/d(textures) against the reference's K4+K5+K6 gradients pushed through the
    gather's adjoint."""
    from deep3dmap_amd import neural_renderer as nr, synthetic
    from deep3dmap_amd.neural_renderer import mesh_ops
    from deep3dmap_amd.neural_renderer.rasterize import rasterize_lit
    eyes_np = synthetic.camera_ring(32)[[1, 4, 8, 12, 15, 19, 23, 27, 30]] if views == 9 else synthetic.camera_ring(32)[3:3 + views]
    v, tri, sv, faces, S = _config_scene(225, eyes_np, 512, False)
    B = len(eyes_np)
    tex0 = torch.from_numpy(synthetic.random_textures(tri.shape[0], 2)).cuda()
    vt = torch.from_numpy(v).cuda()[None]
    ft = torch.from_numpy(tri).cuda()[None]
    light_cfg = (0.5, 0.5, [1, 1, 1], [1, 1, 1], [0, 1, 0])
    # reference side: materialised fill_back textures * light (product's own lighting operator, pinned by
    # test_gather_and_lighting_match_reference_modules), then the reference kernels
    tex_fb = torch.cat((tex0[None], tex0[None].permute(0, 1, 4, 3, 2, 5)), dim=1)
    faces_world = mesh_ops.gather_faces(vt, ft, True)
    tex_lit = mesh_ops.lighting(faces_world, tex_fb.clone(), *light_cfg).expand(B, -1, -1, -1, -1, -1).contiguous()
    ref = RH.forward(faces, tex_lit, S, 0.1, 100.0, 1e-3, (0, 0, 0))
    svg = sv.detach().clone().requires_grad_(True)
    texg = tex0[None].clone().requires_grad_(True)
    from deep3dmap_amd import _lib
    with _lib.coverage_form(form), kernels_launched() as k:
        out = rasterize_lit(svg, vt, ft, texg, light_cfg, True, 512, False, 0.1, 100.0, 1e-3, (0, 0, 0))
    assert_coverage_form_ran(k.names, ran)
    # images: [B,3,s,s] flipped CHW of the internal maps (NR/rasterize.py:305-317)
    assert torch.equal(out["rgb"], ref["rgb_map"].permute(0, 3, 1, 2).flip(2))
    assert torch.equal(out["alpha"], ref["alpha_map"].flip(1))
    assert torch.equal(out["depth"], ref["depth_map"].flip(1))
    gen = torch.Generator(device="cuda").manual_seed(44)
    g_rgb = torch.randn(B, 3, S, S, device="cuda", generator=gen)
    g_alpha = torch.randn(B, S, S, device="cuda", generator=gen)
    g_depth = torch.randn(B, S, S, device="cuda", generator=gen)
    (out["rgb"] * g_rgb).sum().add((out["alpha"] * g_alpha).sum()).add((out["depth"] * g_depth).sum()).backward()
    gf_ref, gt_ref = RH.backward(ref, g_rgb.flip(2).permute(0, 2, 3, 1).contiguous(), g_alpha.flip(1).contiguous(),
                                 g_depth.flip(1).contiguous(), True, True, True)
    # adjoint of vertices_to_faces + fill_back (NR/vertices_to_faces.py:16-22, NR/renderer.py:86): scatter-add
    Ft = tri.shape[0]
    idx = torch.from_numpy(tri).cuda().long()
    idx_fb = torch.cat((idx, idx.flip(-1)), 0).reshape(-1)
    gsv_ref = torch.zeros_like(sv)
    gsv_ref.index_add_(1, idx_fb, gf_ref.reshape(B, 2 * Ft * 3, 3))
    assert _rel_max(svg.grad, gsv_ref) <= GRAD_RTOL, _rel_max(svg.grad, gsv_ref)
    # adjoint of cat(textures, permuted) * light, summed over the views that share the textures
    light = (tex_lit[0] / tex_fb[0].clamp_min(1e-20))                         # recovered per-texel light factor
    gt_fb = (gt_ref * light[None]).sum(0)
    gt_ref0 = gt_fb[:Ft] + gt_fb[Ft:].permute(0, 3, 2, 1, 4)
    assert _rel_max(texg.grad[0], gt_ref0) <= GRAD_RTOL, _rel_max(texg.grad[0], gt_ref0)


def test_config4_pooled_objective_at_full_size_against_reference():
    """the anti-aliased form of the fused objective (k_render_lit_fit_records_pooled: objective on the
    2x2-pooled images, walk records per INTERNAL pixel) had only been checked at <= 64^2.  Here at the headline's size --
    config 4's mesh, output 512^2 = internal 1024^2, three cameras -- against the reference's kernels: the pooled images the
    same pass writes (images_out) equal the reference's maps flipped and pooled (NR/rasterize.py:305-326); the objective's
    value equals the composition of deep3dmap's losses on them; and d(loss)/d(screen vertices), d(loss)/d(textures) equal
    the reference's K4 + K5 + K6 fed with the objective's gradient images spread over the internal pixels (the pooling's
    adjoint: a quarter each), pushed through the gather's adjoint."""
    import torch.nn.functional as Fnn
    from deep3dmap_amd import synthetic
    from deep3dmap_amd.neural_renderer import mesh_ops
    from deep3dmap_amd.neural_renderer.rasterize import rasterize_lit_fit
    s_out = 512
    eyes_np = synthetic.camera_ring(32)[[2, 13, 24]]
    v, tri, sv, faces, S = _config_scene(225, eyes_np, s_out, True)
    assert S == 1024
    B = len(eyes_np)
    tex0 = torch.from_numpy(synthetic.random_textures(tri.shape[0], 2)).cuda()
    vt = torch.from_numpy(v).cuda()[None]
    ft = torch.from_numpy(tri).cuda()[None]
    light_cfg = (0.5, 0.5, [1, 1, 1], [1, 1, 1], [0, 1, 0])
    tex_fb = torch.cat((tex0[None], tex0[None].permute(0, 1, 4, 3, 2, 5)), dim=1)
    faces_world = mesh_ops.gather_faces(vt, ft, True)
    tex_lit = mesh_ops.lighting(faces_world, tex_fb.clone(), *light_cfg).expand(B, -1, -1, -1, -1, -1).contiguous()
    ref = RH.forward(faces, tex_lit, S, 0.1, 100.0, 1e-3, (0, 0, 0))
    # the reference's output images: flip, CHW, 2x2 mean (NR/rasterize.py:305-326) -- leaves of a torch graph
    pool = lambda x: Fnn.avg_pool2d(x, 2)
    rgb_r = pool(ref["rgb_map"].permute(0, 3, 1, 2).flip(2)).requires_grad_(True)
    alpha_r = pool(ref["alpha_map"].flip(1)[:, None])[:, 0].requires_grad_(True)
    depth_r = pool(ref["depth_map"].flip(1)[:, None])[:, 0].requires_grad_(True)
    gen = torch.Generator(device="cuda").manual_seed(45)
    rgb_t = torch.rand(B, 3, s_out, s_out, device="cuda", generator=gen)
    depth_t = (depth_r.detach() + 0.05 * torch.randn(B, s_out, s_out, device="cuda", generator=gen)).contiguous()
    blobs = Fnn.avg_pool2d(torch.rand(B, 1, s_out, s_out, device="cuda", generator=gen), 31, stride=1, padding=15)[:, 0]
    alpha_t = (blobs > blobs.median()).float().contiguous()
    mask = alpha_t
    # deep3dmap's objective on those images (utils.py:105-114 masked L1 means; examples/example2.py:46 squared error)
    den = mask.sum()
    loss_r = ((rgb_r - rgb_t).abs() * mask[:, None]).sum() / (3 * den) + ((alpha_r - alpha_t) ** 2).sum() / (s_out * s_out) + \
        ((depth_r - depth_t).abs() * mask).sum() / den
    loss_r.backward()
    # the pooling's adjoint: every internal pixel of an output pixel gets a quarter of its gradient; un-flip, HWC
    spread = lambda g: (g.repeat_interleave(2, -2).repeat_interleave(2, -1) * 0.25)
    g_rgb = spread(rgb_r.grad).flip(2).permute(0, 2, 3, 1).contiguous()
    g_alpha = spread(alpha_r.grad).flip(1).contiguous()
    g_depth = spread(depth_r.grad).flip(1).contiguous()
    gf_ref, gt_ref = RH.backward(ref, g_rgb, g_alpha, g_depth, True, True, True)
    # product: the fused objective with anti-aliasing, images as a by-product
    svg = sv.detach().clone().requires_grad_(True)
    texg = tex0[None].clone().requires_grad_(True)
    images = (torch.empty(B, 3, s_out, s_out, device="cuda"), torch.empty(B, s_out, s_out, device="cuda"),
              torch.empty(B, s_out, s_out, device="cuda"))
    with kernels_launched() as k:
        loss = rasterize_lit_fit(svg, vt, ft, texg, light_cfg, True, (rgb_t, depth_t, alpha_t, mask), s_out, 0.1, 100.0, 1e-3,
                                 (0, 0, 0), images_out=images, anti_aliasing=True)
        loss.backward()
    assert "k_render_lit_fit_records" in k.names and "k_pack_maps" not in k.names, sorted(k.names)
    rgb_p, depth_p, alpha_p = images
    assert torch.equal(alpha_p, alpha_r.detach())
    assert float((rgb_p - rgb_r.detach()).abs().max()) <= 1e-6 and float((depth_p - depth_r.detach()).abs().max()) <= 1e-5
    assert abs(float(loss) - float(loss_r)) <= 2e-6 * abs(float(loss_r)), (float(loss), float(loss_r))
    Ft = tri.shape[0]
    idx = torch.from_numpy(tri).cuda().long()
    idx_fb = torch.cat((idx, idx.flip(-1)), 0).reshape(-1)
    gsv_ref = torch.zeros_like(sv)
    gsv_ref.index_add_(1, idx_fb, gf_ref.reshape(B, 2 * Ft * 3, 3))
    light = (tex_lit[0] / tex_fb[0].clamp_min(1e-20))
    gt_fb = (gt_ref * light[None]).sum(0)
    gt_ref0 = gt_fb[:Ft] + gt_fb[Ft:].permute(0, 3, 2, 1, 4)
    errs = {"grad_screen_vertices": _rel_max(svg.grad, gsv_ref), "grad_textures": _rel_max(texg.grad[0], gt_ref0),
            "loss_rel": abs(float(loss) - float(loss_r)) / abs(float(loss_r))}
    _record("config4_100352tri_512aa_3views_pooled_objective", {**errs, "pixels_internal": B * S * S, "views": B})
    assert errs["grad_screen_vertices"] <= GRAD_RTOL and errs["grad_textures"] <= GRAD_RTOL, errs
