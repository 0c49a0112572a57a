"""GPU tests of Pt3dRenderer.sample (deep3dmap/core/renderer/renderer_pt3d.py:46-98) on the native rasterizer.
pytorch3d -- whose arithmetic the reference delegates to -- is not available (parity unpinned).  The checks:
a planar quad with known image coordinates in closed form, the triangle filter's literal behaviour, and a DIFFERENTIAL
fuzz against oracle/pt3d_oracle.py, an independent brute-force f64 restatement of the pytorch3d pipeline the reference runs
(values and, through its autograd, the gradients wrt the image and the per-vertex image coordinates)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _quad(n=5):
    """(n x n)-vertex flat grid over [-1, 1]^2 at z = 1, wound to face +z; UV = position (as imgs2mesh.py:46-48 builds
    template_uvs3d from the template's uv coordinates)."""
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, n), torch.linspace(-1, 1, n), indexing="ij")
    verts = torch.stack((xs, ys, torch.ones_like(xs)), -1).reshape(-1, 3)
    idx = torch.arange(n * n).reshape(n, n)
    a, b, c, d = idx[:-1, :-1], idx[:-1, 1:], idx[1:, :-1], idx[1:, 1:]       # (y,x), (y,x+1), (y+1,x), (y+1,x+1)
    tri = torch.cat([torch.stack((a, b, c), -1).reshape(-1, 3), torch.stack((b, d, c), -1).reshape(-1, 3)], 0).int()
    normals = torch.tensor([0., 0., 1.]).repeat(n * n, 1)
    return verts.cuda(), tri.cuda(), normals.cuda()


def _renderer(T):
    from deep3dmap_amd.core import Pt3dRenderer
    return Pt3dRenderer("cuda", T, lookview=torch.tensor([0., 0., 1.]).cuda())


def _expected(imgs, T, affine=(1.0, 0.0, 1.0, 0.0)):
    """closed form on the flat quad at z = 1 facing +z: pixel (row i from the top, col j) sees world (x, y) = ((2j+1)/T - 1,
    1 - (2i+1)/T); its image coordinate is affine in (x, y); bilinear lookup as TexturesUV does.  The light pytorch3d falls
    back to sits at (0, 1, 0), BEHIND that plane (n . l = -1/|.| < 0): diffuse and specular vanish, the ambient 0.5 stays."""
    j = (2 * torch.arange(T, dtype=torch.float32) + 1) / T - 1
    x = j[None, :].expand(T, T)
    y = (-j)[:, None].expand(T, T)
    su, ou, sv, ov = affine
    u, v = su * (x + 1) / 2 + ou, sv * (y + 1) / 2 + ov
    grid = torch.stack((2 * u - 1, 1 - 2 * v), -1)[None].expand(imgs.shape[0], T, T, 2)     # flipped map <=> y -> -y
    tex = F.grid_sample(imgs.cpu(), grid, mode="bilinear", padding_mode="border", align_corners=True)
    shade = torch.full((T, T), 0.5)
    return tex.permute(0, 2, 3, 1) * shade[None, :, :, None], shade


def _oracle(normals, angles, tri, imgs, verts, fp, T, light=None):
    from oracle import pt3d_oracle as P
    return P.sample(normals.cpu(), angles.cpu(), tri.cpu(), imgs, verts.cpu(), fp, torch.tensor([0., 0., 1.]), T, light)


def test_planar_quad_resamples_the_image_and_shades_it():
    T, B = 48, 2
    verts, tri, normals = _quad()
    r = _renderer(T)
    imgs = torch.rand(B, 3, 20, 28).cuda()
    # image coordinates: the quad's own uv in the first batch entry, a shifted / scaled window in the second
    uv0 = (verts[:, :2] + 1) / 2
    uv1 = torch.stack((0.6 * uv0[:, 0] + 0.3, 0.5 * uv0[:, 1] + 0.1), 1)
    face_project = torch.stack((uv0, uv1), 0)
    angles = torch.tensor([[0.2, math.pi - 0.3, 0.1], [0.1, math.pi + 0.2, 0.0]]).cuda()     # turned away: vertices invisible
    img, mask = r.sample(normals, angles, tri, imgs, verts, face_project)
    assert img.shape == (B, T, T, 4) and mask.shape == (B, T, T, 4)
    assert torch.all(img[..., 3] == 1) and torch.all(mask[..., 3] == 1)       # the quad fills the whole UV square
    for b, affine in ((0, (1.0, 0.0, 1.0, 0.0)), (1, (0.6, 0.3, 0.5, 0.1))):
        exp, shade = _expected(imgs[b:b + 1], T, affine)
        assert torch.allclose(img[b, :, :, :3].cpu(), exp[0], atol=2e-5), float((img[b, :, :, :3].cpu() - exp[0]).abs().max())
        assert torch.allclose(mask[b, :, :, :3].cpu(), shade[:, :, None].expand(T, T, 3), atol=2e-6)


def _random_layout(n, seed, z_amp=0.3, flip_fraction=0.3):
    """a jittered n x n UV layout with bumpy z, some triangles wound the other way (pytorch3d does not cull), random unit
    normals for the visibility test, random image coordinates"""
    g = torch.Generator().manual_seed(seed)
    ys, xs = torch.meshgrid(torch.linspace(-0.85, 0.85, n), torch.linspace(-0.85, 0.85, n), indexing="ij")
    jit = 0.6 / (n - 1) * (torch.rand(n, n, 2, generator=g) - 0.5)
    verts = torch.stack((xs + jit[..., 0], ys + jit[..., 1], z_amp * torch.randn(n, n, generator=g)), -1).reshape(-1, 3)
    idx = torch.arange(n * n).reshape(n, n)
    a, b, c, d = idx[:-1, :-1], idx[:-1, 1:], idx[1:, :-1], idx[1:, 1:]
    tri = torch.cat([torch.stack((a, b, c), -1).reshape(-1, 3), torch.stack((b, d, c), -1).reshape(-1, 3)], 0)
    flip = torch.rand(tri.shape[0], generator=g) < flip_fraction
    tri = torch.where(flip[:, None], tri.flip(1), tri).int()
    normals = torch.nn.functional.normalize(torch.randn(n * n, 3, generator=g), dim=1)
    return verts, tri, normals, g


@pytest.mark.parametrize("seed,n,T,C", [(0, 6, 40, 3), (1, 9, 64, 3), (2, 5, 33, 1), (3, 12, 96, 3)])
def test_sample_against_the_independent_oracle_on_random_layouts(seed, n, T, C):
    """differential fuzz against oracle/pt3d_oracle.py (brute-force f64 restatement of the pytorch3d pipeline the
    reference runs: strict-inside hard raster, TexturesUV bilinear, default Phong light): images to f32 noise, gradients wrt the
    image and the per-vertex image coordinates to 1e-3 of their scale"""
    verts, tri, normals, g = _random_layout(n, seed)
    B = 3
    imgs0 = torch.rand(B, C, 18, 22, generator=g)
    fp0 = torch.rand(B, n * n, 2, generator=g) * 0.9 + 0.05
    fp0[1] = fp0[1] * 1.6 - 0.3                                           # partly outside the image: border padding
    angles = torch.tensor([[0.3, 2.6, -0.2], [0.1, -0.4, 0.5], [0.0, math.pi, 0.0]])
    w = torch.randn(B, T, T, 4, generator=g)
    r = _renderer(T)
    imgs, fp = imgs0.clone().cuda().requires_grad_(True), fp0.clone().cuda().requires_grad_(True)
    img, mask = r.sample(normals.cuda(), angles.cuda(), tri.cuda(), imgs, verts.cuda(), fp)
    (img * w.cuda()).sum().backward()
    imgs_o, fp_o = imgs0.clone().double().requires_grad_(True), fp0.clone().double().requires_grad_(True)
    img_o, mask_o = _oracle(normals, angles, tri, imgs_o, verts, fp_o, T)
    (img_o * w.double()).sum().backward()
    assert float(mask_o[..., 3].mean()) > 0.3                             # the layout covers a good part of the square
    rel = lambda a, b_: float((a.cpu().double() - b_).abs().max() / b_.abs().max())
    err = dict(coverage=int((img[..., 3].cpu().double() != img_o[..., 3]).sum()),      # no texel centre sits on an edge
               img=float((img.detach().cpu().double() - img_o.detach()).abs().max()),
               mask=float((mask.cpu().double() - mask_o.detach()).abs().max()),
               g_img=rel(imgs.grad, imgs_o.grad), g_uv=rel(fp.grad, fp_o.grad))
    # values: f32 against f64.  The barycentrics come from the rasterizer's f32 pixel-space face inverse (KCU:44-62:
    # w = a x + b y + c with |c| ~ T / edge length, i.e. a few 1e-6 of cancellation error at T = 96), the texture
    # coordinate is their mix scaled by the image width, the texture's slope is ~1 per texel: ~1e-4 on the colour
    tol = dict(coverage=0, img=3e-4, mask=3e-4, g_img=1e-3, g_uv=1e-3)      # (mask: the ^64 of the specular lobe amplifies too)
    assert all(err[k] <= tol[k] for k in tol), err


def test_light_parameters_and_overlapping_sheets_against_the_oracle():
    """(a) the reference's own (unused) light -- PointLights((0,0,10), ambient 0, diffuse 1, specular 0) -- through the class
    attributes; (b) a lit configuration with a visible specular lobe; (c) two sheets on top of each other: the nearer wins"""
    from deep3dmap_amd.core import Pt3dRenderer
    T = 48
    verts, tri, normals, g = _random_layout(7, 11, z_amp=0.05, flip_fraction=0.0)
    imgs = torch.rand(1, 3, 16, 16, generator=g)
    fp = torch.rand(1, 49, 2, generator=g)
    angles = torch.tensor([[0.0, math.pi, 0.0]])
    cases = [dict(location=(0.0, 0.0, 10.0), ambient=0.0, diffuse=1.0, specular=0.0, shininess=64.0),
             dict(location=(0.3, 0.4, 3.0), ambient=0.2, diffuse=0.5, specular=0.9, shininess=8.0)]
    for lt in cases:
        r = _renderer(T)
        r.LIGHT_LOCATION, r.LIGHT_AMBIENT, r.LIGHT_DIFFUSE = lt["location"], lt["ambient"], lt["diffuse"]
        r.LIGHT_SPECULAR, r.SHININESS = lt["specular"], lt["shininess"]
        img, mask = r.sample(normals.cuda(), angles.cuda(), tri.cuda(), imgs.cuda(), verts.cuda(), fp.cuda())
        img_o, mask_o = _oracle(normals, angles, tri, imgs, verts, fp, T, lt)
        assert float((img.cpu().double() - img_o).abs().max()) < 2e-5 and float((mask.cpu().double() - mask_o).abs().max()) < 2e-5
    assert float(mask_o[..., :3].max()) > float(mask_o[..., :3][mask_o[..., 3] > 0].min()) + 0.05     # shading varies
    # (c) a second copy of the sheet further from the camera (smaller z), listed FIRST: hidden everywhere
    far = verts.clone()
    far[:, 2] -= 0.8
    v2 = torch.cat([far, verts], 0)
    tri2 = torch.cat([tri, tri + 49], 0)
    fp2 = torch.cat([torch.zeros(1, 49, 2), fp], 1)                        # the far sheet would sample the image's corner
    n2 = torch.cat([normals, normals], 0)
    r = _renderer(T)
    img, _ = r.sample(n2.cuda(), angles.cuda(), tri2.cuda(), imgs.cuda(), v2.cuda(), fp2.cuda())
    img_o, _ = _oracle(n2, angles, tri2, imgs, v2, fp2, T)
    assert float((img.cpu().double() - img_o).abs().max()) < 1e-5


def test_triangle_filter_as_written_and_partial_coverage():
    """renderer_pt3d.py:59-74 keeps every triangle once ANY vertex is invisible and none when all are visible."""
    T = 32
    verts, tri, normals = _quad()
    r = _renderer(T)
    imgs = torch.rand(2, 3, 16, 16).cuda()
    fp = ((verts[:, :2] + 1) / 2)[None].repeat(2, 1, 1)
    angles = torch.tensor([[0.0, 0.0, 0.0], [0.0, math.pi, 0.0]]).cuda()       # facing the view / turned away
    img, mask = r.sample(normals, angles, tri, imgs, verts, fp)
    assert torch.all(img[0] == 0) and torch.all(mask[0] == 0)                  # all vertices visible -> no faces kept
    assert torch.all(mask[1, :, :, 3] == 1)
    # one vertex normal flipped: that vertex is invisible at angle 0 -> every triangle is kept
    n2 = normals.clone()
    n2[7] = -n2[7]
    img2, _ = r.sample(n2, angles[:1], tri, imgs[:1], verts, fp[:1])
    assert torch.allclose(img2[0], img[1] * 0 + r.sample(normals, angles[1:], tri, imgs[:1], verts, fp[:1])[0][0])
    # a mesh that covers only part of the UV square leaves the rest at the background (0) with alpha 0
    half = verts.clone()
    half[:, 0] = half[:, 0] * 0.5 - 0.5                                         # x in [-1, 0]
    img3, mask3 = r.sample(normals, angles[1:], tri, imgs[:1], half, fp[:1])
    assert torch.all(mask3[0, :, T // 2 + 1:, :] == 0) and torch.all(mask3[0, :, :T // 2 - 1, 3] == 1)


def test_reversed_winding_is_rendered_and_lit_from_its_own_side():
    """pytorch3d rasterizes both windings (no culling); reversing the winding flips the vertex normals, so the plane at
    z = 1 now faces the default light at (0, 1, 0) and picks up the diffuse term on top of the ambient one"""
    T = 24
    verts, tri, normals, _ = _random_layout(4, 21, z_amp=0.0, flip_fraction=1.0)      # every triangle wound clockwise
    verts[:, 2] = 1.0
    r = _renderer(T)
    imgs = torch.ones(1, 3, 8, 8)
    fp = ((verts[:, :2] + 1) / 2)[None]
    angles = torch.tensor([[0.0, math.pi, 0.0]])
    img, mask = r.sample(normals.cuda(), angles.cuda(), tri.cuda(), imgs.cuda(), verts.cuda(), fp.cuda())
    img_o, mask_o = _oracle(normals, angles, tri, imgs, verts, fp, T)
    covered = mask[0, :, :, 3] == 1
    assert float(covered.float().mean()) > 0.5 and float(mask[0, :, :, :3][covered].min()) > 0.5 + 0.05
    assert float((img.cpu().double() - img_o).abs().max()) < 1e-5 and float((mask.cpu().double() - mask_o).abs().max()) < 1e-5


def test_euler_matrix_and_vertex_normals_helpers():
    from deep3dmap_amd.core.renderer_pt3d import euler_xyz_to_matrix, vertex_normals
    a = torch.tensor([[0.3, -0.2, 0.5]])
    R = euler_xyz_to_matrix(a)[0]
    cx, sx, cy, sy, cz, sz = math.cos(.3), math.sin(.3), math.cos(-.2), math.sin(-.2), math.cos(.5), math.sin(.5)
    Rx = torch.tensor([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = torch.tensor([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    assert torch.allclose(R, (Rx @ Ry @ Rz).float(), atol=1e-6)
    verts, tri, _ = _quad(4)
    assert torch.allclose(vertex_normals(verts, tri), torch.tensor([0., 0., 1.]).cuda().expand(16, 3), atol=1e-6)
