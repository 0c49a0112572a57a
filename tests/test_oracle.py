"""CPU tests: the oracle (oracle/nr_oracle.c + nr_oracle.py) against the committed golden vectors -- kern/* made by the
reference's own kernels run on the device (tests/golden/make_golden_kern.py), the rest by its imported pure-torch
modules (tests/golden/make_golden.py) -- and against the reference's own known-answer tests.  (Fresh random scenes,
port vs the device reference: tests/test_gpu_reference.py.)"""
import numpy as np
import pytest
import torch

from conftest import golden_case, golden_case_names
from oracle import nr_oracle as O

CASES = ["rand_b2_f24_s32_ts2", "rand_b1_f64_s64_ts4", "rand_b3_f12_s16_ts1", "nearfar_b1_f32_s32",
         "ties_zero_batch_b2_f32_s32", "fillback_small_b2_f80_s48", "grid_on_pixel_centres_b1_f128_s32"]


def test_case_list_matches_fixture(golden):
    assert golden_case_names(golden) == sorted(CASES)


def _forward(c, bbox=False):
    return O.raster_forward(c["faces"], c["textures"], int(c["image_size"]), float(c["near"]), float(c["far"]),
                            float(c["eps"]), c["background"], True, True, True, backend="port", bbox=bbox)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("bbox", [False, True])
def test_forward_maps_bit_exact(golden, name, bbox):
    c = golden_case(golden, name)
    m = _forward(c, bbox)
    # integer/index work: bit exact; float maps: same IEEE operations in the same order -> bit exact too
    for key in ("face_index_map", "sampling_index_map"):
        assert np.array_equal(m[key], c[key]), key
    for key in ("faces_inv", "weight_map", "depth_map", "face_inv_map", "rgb_map", "alpha_map", "sampling_weight_map"):
        assert np.array_equal(m[key], c[key], equal_nan=True), key


@pytest.mark.parametrize("name", CASES)
def test_backward_kernels(golden, name):
    c = golden_case(golden, name)
    m = _forward(c)
    # The vectors come from the device: K5 / K6 add with float atomics there (order of additions not fixed), K4 is
    # one thread per face.  Hence: same sums up to a few ulps of the tensor's scale, not bit patterns.
    def close(got, ref):
        assert np.array_equal(np.isfinite(got), np.isfinite(ref))
        ok = np.isfinite(ref)
        return np.abs(got[ok] - ref[ok]).max() <= 2e-6 * max(np.abs(ref[ok]).max(), 1e-30)
    gf, gt = O.raster_backward(m, c["grad_rgb_map"], c["grad_alpha_map"], c["grad_depth_map"], True, True, True)
    assert close(gf, c["grad_faces_all"])
    assert close(gt, c["grad_textures"])
    gf, _ = O.raster_backward(m, c["grad_rgb_map"], c["grad_alpha_map"], None, True, True, False)
    assert close(gf, c["grad_faces_pixel_rgba"])
    gf, _ = O.raster_backward(m, None, c["grad_alpha_map"], None, False, True, False)
    assert close(gf, c["grad_faces_pixel_alpha"])
    gf, _ = O.raster_backward(m, None, None, c["grad_depth_map"], False, False, True)
    assert close(gf, c["grad_faces_depth"])


def test_ties_pick_lowest_index_and_zero_batch_is_empty(golden):
    c = golden_case(golden, "ties_zero_batch_b2_f32_s32")
    fi = c["face_index_map"]
    assert fi[0].max() < 16          # duplicates live at f+16 and must never win
    assert (fi[1] == -1).all()       # all-zero vertices: NaN weights, never selected (KCU:127-139)
    assert (c["depth_map"][1] == c["far"]).all()


def _known(golden, which, mode):
    v = golden[f"known/{which}/vertices"]
    pxi, pyi = (int(t) for t in golden[f"known/{which}/pxy"])
    B = 4                                                  # tests/utils.py:11-21 to_minibatch: slot 2 of 4
    vert = torch.zeros(B, 3, 3)
    vert[2] = torch.from_numpy(v)
    faces = torch.zeros(B, 1, 3, dtype=torch.int32)
    faces[2] = torch.tensor([[0, 1, 2]], dtype=torch.int32)
    vert.requires_grad = True
    r = O.Renderer(camera_mode="look_at")
    r.image_size, r.anti_aliasing, r.perspective = 64, False, False
    if mode == "sil":
        im = r(vert, faces, mode="silhouettes")
    else:
        r.light_intensity_ambient, r.light_intensity_directional = 1.0, 0.0
        tex = torch.zeros(B, 1, 4, 4, 4, 3)
        tex[2] = 1
        im = r(vert, faces, tex)[0].mean(1)
    target = 1 if which == "sil1" else 0
    torch.sum(torch.abs(im[:, pyi, pxi] - target)).backward()
    return vert.grad.numpy(), golden[f"known/{which}/grad"]


@pytest.mark.parametrize("which", ["sil1", "sil2"])
@pytest.mark.parametrize("mode", ["sil", "rgb"])
def test_reference_known_answer_gradients(golden, which, mode):
    """tests/test_rasterize_silhouettes.py:37-99 and tests/test_rasterize.py:84-156 (rtol 1e-2)."""
    g, ref = _known(golden, which, mode)
    assert np.allclose(g[2], ref, rtol=1e-2, atol=1e-8)
    assert np.abs(g[[0, 1, 3]]).max() == 0


def test_look_at_and_perspective_known_answers(golden):
    """tests/test_look_at.py:9-25, tests/test_perspective.py:9-14."""
    v = torch.tensor([[[1., 0., 0.]]])
    for eye, ans in zip(golden["known/look_at/eyes"], golden["known/look_at/answers"]):
        assert np.allclose(O.look_at(v, eye).squeeze().numpy(), ans)
    out = O.perspective(torch.from_numpy(golden["known/perspective/in"])[None, None])
    assert np.allclose(out.squeeze().numpy(), golden["known/perspective/out"])


def test_camera_restatements_match_reference_modules(golden):
    g = golden
    t = lambda k: torch.from_numpy(g[k])
    v = t("cam/projection/vertices").requires_grad_(True)
    out = O.projection(v, t("cam/projection/K"), t("cam/projection/R"), t("cam/projection/t"),
                       t("cam/projection/dist"), int(g["cam/projection/orig_size"]))
    assert torch.allclose(out, t("cam/projection/out"), rtol=1e-6, atol=1e-7)
    out.backward(t("cam/projection/grad_out"))
    assert torch.allclose(v.grad, t("cam/projection/grad_vertices"), rtol=1e-5, atol=1e-6)
    v = t("cam/look_at/vertices").requires_grad_(True)
    la = O.look_at(v, t("cam/look_at/eyes"))
    assert torch.allclose(la, t("cam/look_at/out"), rtol=1e-6, atol=1e-7)
    pp = O.perspective(la, float(g["cam/look_at/persp_angle"]))
    assert torch.allclose(pp, t("cam/look_at/persp_out"), rtol=1e-6, atol=1e-7)
    pp.backward(t("cam/look_at/grad_out"))
    assert torch.allclose(v.grad, t("cam/look_at/grad_vertices"), rtol=1e-5, atol=1e-6)
    assert torch.allclose(O.look_at(t("cam/look_at/vertices"), [0, 0, -2.732]), t("cam/look_at/single_eye_out"),
                          rtol=1e-6, atol=1e-7)
    for (d, e, a), ref in zip(g["cam/points_from_angles/in"], g["cam/points_from_angles/out"]):
        assert np.allclose(O.get_points_from_angles(float(d), float(e), float(a)), ref, rtol=1e-6)
    f = O.vertices_to_faces(t("cam/projection/vertices"), t("cam/v2f/faces"))
    assert torch.equal(f, t("cam/v2f/out"))
    p = g["cam/lighting/params"]
    lit = O.lighting(f, t("cam/lighting/textures"), float(p[0]), float(p[1]), tuple(p[2:5]), tuple(p[5:8]), tuple(p[8:11]))
    assert torch.allclose(lit, t("cam/lighting/out"), rtol=1e-6, atol=1e-7)


def test_deep3dmap_helpers_and_losses(golden):
    g = golden
    t = lambda k: torch.from_numpy(g[k])
    rot, tr = O.get_transform_matrices(t("d3m/view6"))
    assert torch.allclose(rot, t("d3m/rot"), atol=1e-7) and torch.equal(tr, t("d3m/trans"))
    rot5, tr5 = O.get_transform_matrices(t("d3m/view6")[:, :5])
    assert torch.allclose(rot5, t("d3m/rot5"), atol=1e-7) and torch.equal(tr5, t("d3m/trans5"))
    assert torch.equal(O.get_grid(2, 3, 4, normalize=False), t("d3m/grid_2_3_4_raw"))
    assert torch.allclose(O.get_grid(1, 3, 4, normalize=True), t("d3m/grid_1_3_4_norm"))
    assert torch.equal(O.get_face_idx(2, 3, 4), t("d3m/face_idx_2_3_4"))
    assert torch.allclose(O.get_textures_from_im(t("d3m/im"), 1), t("d3m/tex_from_im_ts1"))
    assert torch.allclose(O.get_textures_from_im(t("d3m/im"), 2), t("d3m/tex_from_im_ts2"), atol=1e-7)
    a, b, m, s = t("loss/a"), t("loss/b"), t("loss/mask"), t("loss/sigma")
    assert torch.allclose(O.photometric_loss(a, b), t("loss/photometric"))
    assert torch.allclose(O.photometric_loss(a, b, mask=m), t("loss/photometric_mask"))
    assert torch.allclose(O.photometric_loss(a, b, mask=m, conf_sigma=s), t("loss/photometric_mask_sigma"))
    assert torch.allclose(O.smooth_loss(a[:, 0]), t("loss/smooth"))
    assert torch.allclose(O.smooth_loss([a[:, 0], b[:, 0, ::2, ::2]]), t("loss/smooth_pyramid"))
