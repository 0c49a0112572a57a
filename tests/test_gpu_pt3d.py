"""GPU tests of Pt3dRenderer.sample (deep3dmap/core/renderer/renderer_pt3d.py:46-98) on the native rasterizer.
pytorch3d -- whose arithmetic the reference delegates to -- is not available (parity unpinned), so the
checks are analytic: a planar quad with known image coordinates, the closed form of the diffuse term, the triangle
filter's literal behaviour, and the adjoint against torch autograd of the same formulas."""
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
    """closed form on the flat quad: pixel (row i from the top, col j) sees world (x, y) = ((2j+1)/T - 1, 1 - (2i+1)/T);
    its image coordinate is affine in (x, y); bilinear lookup as TexturesUV does; diffuse = cos of the angle between
    +z and the direction to the light at (0, 0, 10) from z = 1."""
    j = (2 * torch.arange(T, dtype=torch.float32) + 1) / T - 1
    x = j[None, :].expand(T, T)
    y = (-j)[:, None].expand(T, T)
    su, ou, sv, ov = affine
    u, v = su * (x + 1) / 2 + ou, sv * (y + 1) / 2 + ov
    grid = torch.stack((2 * u - 1, 1 - 2 * v), -1)[None].expand(imgs.shape[0], T, T, 2)     # flipped map <=> y -> -y
    tex = F.grid_sample(imgs.cpu(), grid, mode="bilinear", padding_mode="border", align_corners=True)
    shade = 9.0 / torch.sqrt(x ** 2 + y ** 2 + 81.0)
    return tex.permute(0, 2, 3, 1) * shade[None, :, :, None], shade


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


def test_reversed_winding_is_rendered_but_unlit():
    """pytorch3d rasterizes both windings (no culling); a face whose normal points away from the light gets diffuse 0."""
    T = 24
    verts, tri, normals = _quad(3)
    r = _renderer(T)
    imgs = torch.ones(1, 3, 8, 8).cuda()
    fp = ((verts[:, :2] + 1) / 2)[None]
    img, mask = r.sample(normals, torch.tensor([[0.0, math.pi, 0.0]]).cuda(), tri.flip(1).contiguous(), imgs, verts, fp)
    assert torch.all(mask[0, :, :, 3] == 1) and torch.all(mask[0, :, :, :3] == 0) and torch.all(img[0, :, :, :3] == 0)


def test_gradients_wrt_image_and_image_coordinates():
    T, B = 24, 2
    verts, tri, normals = _quad(4)
    r = _renderer(T)
    gen = torch.Generator().manual_seed(3)
    imgs0 = torch.rand(B, 3, 12, 14, generator=gen)
    uv_base = (verts[:, :2].cpu() + 1) / 2
    fp0 = (0.7 * uv_base + 0.15)[None].repeat(B, 1, 1) + 0.01 * torch.randn(B, verts.shape[0], 2, generator=gen)
    g = torch.randn(B, T, T, 4, generator=gen)
    angles = torch.tensor([[0.0, math.pi, 0.0]]).repeat(B, 1).cuda()
    imgs = imgs0.clone().cuda().requires_grad_(True)
    fp = fp0.clone().cuda().requires_grad_(True)
    img, _ = r.sample(normals, angles, tri, imgs, verts, fp)
    (img * g.cuda()).sum().backward()
    # the same forward in torch: per-pixel image coordinates from the product's own coverage maps
    fi, wm, _ = r._uv_coverage(verts, tri)
    fi, wm = fi[0].flip(0).cpu().long(), wm[0].flip(0).cpu()                 # to top-down rows
    F_ = tri.shape[0]
    tri_c = tri.cpu().long()
    ids = torch.where((fi >= F_)[..., None], tri_c[fi % F_].flip(-1), tri_c[fi % F_])       # fill_back: reversed corners
    imgs_t, fp_t = imgs0.clone().requires_grad_(True), fp0.clone().requires_grad_(True)
    uv = (fp_t[:, ids] * wm[None, ..., None]).sum(3)                          # [B,T,T,2]
    grid = torch.stack((2 * uv[..., 0] - 1, 1 - 2 * uv[..., 1]), -1)
    tex = F.grid_sample(imgs_t, grid, mode="bilinear", padding_mode="border", align_corners=True).permute(0, 2, 3, 1)
    pos = (verts.cpu()[ids] * wm[..., None]).sum(2)
    d = torch.tensor([0., 0., 10.]) - pos
    shade = (d[..., 2] / d.norm(dim=-1)).clamp(min=0)
    ref = tex * shade[None, ..., None]
    assert torch.allclose(img[..., :3].detach().cpu(), ref.detach(), atol=2e-5)
    (ref * g[..., :3]).sum().backward()
    assert torch.allclose(imgs.grad.cpu(), imgs_t.grad, atol=1e-4, rtol=1e-4)
    assert torch.allclose(fp.grad.cpu(), fp_t.grad, atol=2e-3 * float(fp_t.grad.abs().max()), rtol=1e-3)


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
