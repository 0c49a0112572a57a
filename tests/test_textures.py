"""Texture asset kernels: uv image -> per-face texture cubes (load_textures) and texture cubes ->
atlas image (create_texture_image).  CPU tests pin the C restatement to the reference-generated vectors in
tests/golden/tex_golden.npz (tests/golden/make_golden_textures.py); GPU tests hold the HIP kernels and the OBJ
texture I/O built on them to the same vectors, bit for bit."""
import os

import numpy as np
import pytest
import torch

from oracle import nr_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tex_golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "tex_golden.npz"))


def _cases(g, kind):
    return sorted({k.split("/")[1] for k in g.files if k.startswith(kind + "/")})


def _case(g, kind, name):
    pre = f"{kind}/{name}/"
    return {k[len(pre):]: g[k] for k in g.files if k.startswith(pre)}


LOAD_CASES = [f"w{w}_b{b}" for w in range(4) for b in (0, 1)]
ATLAS_CASES = ["f10_ts2_o8", "f16_ts4_o16", "f17_ts6_o4", "f1_ts3_o5", "f7_ts4_o16"]


def test_fixture_case_lists(tex_golden):
    assert _cases(tex_golden, "load") == sorted(LOAD_CASES)
    assert _cases(tex_golden, "atlas") == sorted(ATLAS_CASES)


# ------------------------------------------------------------------------------------------ CPU: the oracle
@pytest.mark.parametrize("name", LOAD_CASES)
def test_oracle_load_textures_bit_exact(tex_golden, name):
    c = _case(tex_golden, "load", name)
    uv0 = c["faces_uv"].copy()
    out = O.load_textures_np(c["image"], c["faces_uv"], c["textures_in"].copy(), c["is_update"], int(c["wrapping"]),
                             int(c["bilinear"]), backend="port")
    assert np.array_equal(out, c["textures_out"])
    assert np.array_equal(c["faces_uv"], uv0)                       # inputs are not modified
    keep = c["is_update"] == 0
    assert keep.any() and np.array_equal(out[keep], c["textures_in"][keep])
    if int(c["wrapping"]) == 3:                                     # CLAMP_TO_BORDER writes zeros (LTK:97,109)
        assert np.all(out[~keep] == 0)


@pytest.mark.parametrize("name", ATLAS_CASES)
def test_oracle_create_texture_image_bit_exact(tex_golden, name):
    c = _case(tex_golden, "atlas", name)
    image, vertices = O.create_texture_image_np(c["textures"], int(c["tso"]), backend="port")
    valid = c["valid"]
    assert image.shape == c["image"].shape
    assert np.array_equal(image[valid], c["image"][valid])
    assert np.all(image[~valid] == 0)                               # padding tiles: defined as zeros
    assert np.array_equal(vertices, c["vertices"])


def test_oracle_load_textures_known_answers():
    """Hand-checkable cases: a constant image gives constant cubes; a horizontal ramp sampled at the three corner
    texels returns the ramp value at each uv corner (nearest and bilinear agree on pixel centres)."""
    F, ts, H, W = 2, 3, 5, 9
    const = np.full((H, W, 3), 0.25, np.float32)
    uv = np.array([[[0.1, 0.2], [0.8, 0.3], [0.4, 0.9]], [[0.5, 0.5], [0.625, 0.5], [0.5, 0.75]]], np.float32)
    upd = np.ones(F, np.int32)
    out = O.load_textures_np(const, uv, np.zeros((F, ts, ts, ts, 3), np.float32), upd, 2, True)
    assert np.allclose(out, 0.25, atol=1e-6)
    ramp = np.broadcast_to(np.linspace(0, 1, W, dtype=np.float32)[None, :, None], (H, W, 3)).copy()
    for bilinear in (True, False):
        out = O.load_textures_np(ramp, uv, np.zeros((F, ts, ts, ts, 3), np.float32), upd, 2, bilinear)
        # texel (ts-1,0,0) is corner 0, (0,ts-1,0) corner 1, (0,0,ts-1) corner 2 (LTK:42-50)
        corners = np.stack([out[1, ts - 1, 0, 0, 0], out[1, 0, ts - 1, 0, 0], out[1, 0, 0, ts - 1, 0]])
        assert np.allclose(corners, uv[1, :, 0], atol=1e-6)         # face 1's u values sit on pixel centres


def test_oracle_atlas_round_trip():
    """create_texture_image followed by load_textures on the written uv layout reproduces smooth textures: the
    property the reference's save_obj -> load_obj(load_texture=True) pipeline relies on."""
    rng = np.random.default_rng(3)
    F, ts = 6, 4
    base = rng.random((F, 1, 1, 1, 3)).astype(np.float32)
    tex = np.broadcast_to(base, (F, ts, ts, ts, 3)).copy()           # constant colour per face
    image, vertices = O.create_texture_image_np(tex, 16)
    back = O.load_textures_np(image[::-1].copy(), vertices, np.zeros_like(tex), np.ones(F, np.int32), 2, True)
    back[:, 0, 0, 0] = tex[:, 0, 0, 0]      # texel (0,0,0) has no barycentric direction: it samples uv (0,0) (LTK:45)
    assert np.allclose(back, tex, atol=2e-5)


# ------------------------------------------------------------------------------------------ GPU: the HIP kernels
def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("name", LOAD_CASES)
def test_hip_load_textures_bit_exact(tex_golden, name):
    from deep3dmap_amd.neural_renderer import obj_io
    c = _case(tex_golden, "load", name)
    uv = _dev(c["faces_uv"])
    out = obj_io.load_textures_from_image(_dev(c["image"]), uv, _dev(c["textures_in"]), _dev(c["is_update"]),
                                          int(c["wrapping"]), bool(c["bilinear"]))
    assert np.array_equal(out.cpu().numpy(), c["textures_out"])
    assert np.array_equal(uv.cpu().numpy(), c["faces_uv"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ATLAS_CASES)
def test_hip_create_texture_image_bit_exact(tex_golden, name):
    from deep3dmap_amd.neural_renderer import obj_io
    c = _case(tex_golden, "atlas", name)
    image, vertices = obj_io.create_texture_image(_dev(c["textures"]), int(c["tso"]))
    valid = c["valid"]
    assert np.array_equal(image[valid], c["image"][valid])
    assert np.all(image[~valid] == 0)
    assert np.array_equal(vertices, c["vertices"])


@pytest.mark.gpu
def test_hip_texture_kernels_match_oracle_at_asset_scale():
    """A 20k-face mesh with 8^3 cubes and a 1024x768 image, every wrapping mode; then its 16-px atlas."""
    from deep3dmap_amd.neural_renderer import obj_io
    rng = np.random.default_rng(11)
    F, ts, H, W = 20000, 8, 768, 1024
    image = rng.random((H, W, 3)).astype(np.float32)
    uv = rng.uniform(-0.5, 1.5, (F, 3, 2)).astype(np.float32)
    upd = (rng.random(F) > 0.1).astype(np.int32)
    tin = rng.random((F, ts, ts, ts, 3)).astype(np.float32)
    for wrapping in range(4):
        for bilinear in (True, False):
            want = O.load_textures_np(image, uv, tin.copy(), upd, wrapping, bilinear)
            got = obj_io.load_textures_from_image(_dev(image), _dev(uv), _dev(tin), _dev(upd), wrapping, bilinear)
            assert np.array_equal(got.cpu().numpy(), want), (wrapping, bilinear)
    want_img, want_v = O.create_texture_image_np(tin, 16)
    got_img, got_v = obj_io.create_texture_image(_dev(tin), 16)
    assert np.array_equal(got_img, want_img) and np.array_equal(got_v, want_v)


@pytest.mark.gpu
def test_texture_kernel_argument_errors():
    from deep3dmap_amd.neural_renderer import obj_io
    img, uv = torch.rand(4, 4, 3).cuda(), torch.rand(2, 3, 2).cuda()
    tex, upd = torch.zeros(2, 2, 2, 2, 3).cuda(), torch.ones(2, dtype=torch.int32).cuda()
    with pytest.raises(RuntimeError):
        obj_io.load_textures_from_image(img.cpu(), uv, tex, upd)
    with pytest.raises(RuntimeError):
        obj_io.load_textures_from_image(img.permute(1, 0, 2), uv, tex, upd)     # not contiguous
    with pytest.raises(ValueError):
        obj_io.load_textures_from_image(img, uv[:1], tex, upd)
    with pytest.raises(TypeError):
        obj_io.load_textures_from_image(img, uv, tex, upd.long())
    with pytest.raises(RuntimeError):
        obj_io.load_textures_from_image(img, uv, tex, upd, texture_wrapping=7)   # D3M_ERR_INVALID


def _write_textured_quad(tmp_path, with_kd=True):
    """A two-material OBJ: a textured quad (fanned into 2 triangles) and a Kd-coloured triangle."""
    from PIL import Image
    rng = np.random.default_rng(5)
    img = (rng.random((12, 10, 3)) * 255).astype(np.uint8)
    Image.fromarray(img).save(tmp_path / "quad.png")
    (tmp_path / "quad.mtl").write_text(
        "newmtl painted\nKd 0.2 0.4 0.6\nmap_Kd quad.png\n\nnewmtl plain\nKd 0.9 0.1 0.3\n")
    (tmp_path / "quad.obj").write_text(
        "mtllib quad.mtl\n"
        "v -1 -1 0\nv 1 -1 0\nv 1 1 0\nv -1 1 0\nv 0 0 1\n"
        "vt 0.1 0.15\nvt 0.9 0.1\nvt 0.85 0.9\nvt 0.12 0.8\n"
        "usemtl painted\nf 1/1 2/2 3/3 4/4\n"
        "usemtl plain\nf 1//1 2//2 5//5\n")
    return img


@pytest.mark.gpu
def test_load_obj_with_textures_matches_oracle_pipeline(tmp_path):
    import deep3dmap_amd.neural_renderer as nr
    img = _write_textured_quad(tmp_path)
    vertices, faces, textures = nr.load_obj(str(tmp_path / "quad.obj"), load_texture=True, texture_size=4)
    assert tuple(vertices.shape) == (5, 3) and tuple(faces.shape) == (3, 3) and tuple(textures.shape) == (3, 4, 4, 4, 3)
    assert faces.cpu().tolist() == [[0, 1, 2], [0, 2, 3], [0, 1, 4]]
    # the oracle's restatement of NR/load_obj.py:33-98 on the same files
    vt = np.array([[0.1, 0.15], [0.9, 0.1], [0.85, 0.9], [0.12, 0.8]], np.float32)
    uv = np.stack([vt[[0, 1, 2]], vt[[0, 2, 3]], vt[[-1, -1, -1]]])     # faces without vt use index 0 - 1 = -1
    want = np.full((3, 4, 4, 4, 3), 0.5, np.float32)
    want[:2] = np.array([0.2, 0.4, 0.6], np.float32)
    want[2] = np.array([0.9, 0.1, 0.3], np.float32)
    image = (img.astype(np.float32) / 255.)[::-1].copy()
    want = O.load_textures_np(image, uv, want, np.array([1, 1, 0], np.int32), 0, True)
    assert np.array_equal(textures.cpu().numpy(), want)
    with pytest.raises(Exception, match="Failed to load textures"):
        (tmp_path / "nomtl.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
        nr.load_obj(str(tmp_path / "nomtl.obj"), load_texture=True)


@pytest.mark.gpu
def test_save_obj_with_textures_round_trip(tmp_path):
    """save_obj(textures) writes .obj + .mtl + .png; loading it back reproduces geometry exactly and per-face
    constant colours to 8-bit image precision (the reference's test_save_obj.py leaves this case empty)."""
    import deep3dmap_amd.neural_renderer as nr
    rng = np.random.default_rng(8)
    vertices = torch.tensor([[1., 0., 0.], [0., 1., 0.], [0., 0., 1.], [0., 0., 0.]]).cuda()
    faces = torch.tensor([[1, 3, 2], [3, 1, 0], [2, 0, 1], [0, 2, 3]], dtype=torch.int32).cuda()
    colours = rng.random((4, 1, 1, 1, 3)).astype(np.float32)
    textures = torch.from_numpy(np.broadcast_to(colours, (4, 4, 4, 4, 3)).copy()).cuda()
    path = str(tmp_path / "tetra.obj")
    nr.save_obj(path, vertices, faces, textures)
    for ext in (".obj", ".mtl", ".png"):
        assert os.path.exists(path[:-4] + ext)
    v2, f2, t2 = nr.load_obj(path, normalization=False, load_texture=True, texture_size=4,
                             texture_wrapping='CLAMP_TO_EDGE')
    assert torch.allclose(v2, vertices) and torch.equal(f2, faces)
    t2 = t2.cpu().numpy()
    t2[:, 0, 0, 0] = colours[:, 0, 0, 0]    # texel (0,0,0) has no barycentric direction: it samples uv (0,0) (LTK:45)
    assert np.allclose(t2, textures.cpu().numpy(), atol=1.5 / 255)
    # geometry-only path and the reference's tetrahedron expectations (tests/test_load_obj.py:14-36)
    nr.save_obj(path, vertices, faces)
    v3, f3 = nr.load_obj(path, normalization=False)
    assert torch.allclose(v3, vertices) and torch.equal(f3, faces)
    v4, _ = nr.load_obj(path, normalization=True)
    assert torch.allclose(v4, vertices * 2 - 1.0)
    m = nr.Mesh.fromobj(path[:-4] + ".obj", load_texture=False)
    assert m.num_vertices == 4 and m.num_faces == 4 and tuple(m.textures.shape) == (4, 4, 4, 4, 3)
