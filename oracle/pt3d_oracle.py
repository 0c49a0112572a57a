"""TEST INFRASTRUCTURE -- an independent CPU restatement of what deep3dmap's Pt3dRenderer.sample computes
(deep3dmap/core/renderer/renderer_pt3d.py:46-98).  Only tests/ may import it.

PARITY UNPINNED.  The reference delegates this path to pytorch3d (pinned pytorch3d==0.6.1 in requirements/requirements.txt:132;
third party, neither vendored under /root/reference nor installed here, and there is no network), and holds no test or
golden vector for it.  This file therefore restates the library's PUBLISHED algorithm -- the documented behaviour of
MeshRasterizer (naive path), TexturesUV.sample_textures, PointLights / Materials / phong_shading and hard coverage -- by
brute force in float64 torch, sharing NO code with the product (deep3dmap_amd/core/renderer_pt3d.py runs on the tile
rasterizer and d3m_uv_unwrap): it cannot prove agreement with pytorch3d, it does make the product's test a differential
one instead of a self-check.

  pytorch3d pieces restated (0.6.1):
    transforms.euler_angles_to_matrix(a, "XYZ") = Rx(a0) Ry(a1) Rz(a2);  Rotate(R).transform_normals(n) = n R   (R orthonormal)
    look_at_view_transform(2.7, 0, 0) + OpenGLOrthographicCameras (scale 1): the camera sits at (0, 0, 2.7) looking down -z;
        NDC has +x to the LEFT and +y up, so texel (row i from the top, column j) of a T x T image sees world
        x = (2j + 1)/T - 1,  y = 1 - (2i + 1)/T,  and depth 2.7 - z
    MeshRasterizer(blur_radius=0, faces_per_pixel=1, naive): a face covers a texel centre iff its three barycentric
        coordinates are all > 0 (both windings: cull_backfaces is off; faces of |area| <= 1e-8 are skipped); the nearest
        (smallest interpolated depth, >= 0) wins, lower face index among equals
    TexturesUV(maps, faces_uvs, verts_uvs).sample_textures: uv = barycentric mix of the face's verts_uvs;
        grid_sample(flip(maps, H), uv * 2 - 1, bilinear, align_corners=True, padding_mode="border")
    SoftPhongShader built WITHOUT lights / materials (renderer_pt3d.py:87-91 never passes the ones of :42-43):
        PointLights() = location (0,1,0), ambient 0.5, diffuse 0.3, specular 0.2;  Materials() = 1, 1, 1, shininess 64
        colour = (ambient + diffuse relu(n.l)) texel + specular relu(v.r)^shininess [n.l > 0],  n = barycentric mix of the
        vertex normals (Meshes.verts_normals_packed: area-weighted face normals, normalised), l / v towards the light /
        the camera from the barycentric mix of the vertex positions, r = -l + 2 (n.l) n;  F.normalize eps 1e-6
    (a texel centre EXACTLY on an edge is outside by that strict test; the product's rasterizer follows neural_renderer's
    inclusive rule there -- the differential tests use jittered layouts, where no centre lies on an edge)
    the hard part of softmax_rgb_blend at sigma = gamma = 1e-4: rgb of the covering face, background 0; alpha here is the
        hard coverage (the library's is a sigmoid of the distance to the face's edge; the caller reads rgb only)
"""
import math

import torch
import torch.nn.functional as F

DEFAULT_LIGHT = dict(location=(0.0, 1.0, 0.0), ambient=0.5, diffuse=0.3, specular=0.2, shininess=64.0)
CAMERA_DISTANCE = 2.7


def euler_angles_to_matrix_xyz(angles):
    c, s = torch.cos(angles), torch.sin(angles)
    B = angles.shape[0]
    rx, ry, rz = (torch.eye(3, dtype=angles.dtype).repeat(B, 1, 1) for _ in range(3))
    rx[:, 1, 1], rx[:, 1, 2], rx[:, 2, 1], rx[:, 2, 2] = c[:, 0], -s[:, 0], s[:, 0], c[:, 0]
    ry[:, 0, 0], ry[:, 0, 2], ry[:, 2, 0], ry[:, 2, 2] = c[:, 1], s[:, 1], -s[:, 1], c[:, 1]
    rz[:, 0, 0], rz[:, 0, 1], rz[:, 1, 0], rz[:, 1, 1] = c[:, 2], -s[:, 2], s[:, 2], c[:, 2]
    return rx @ ry @ rz


def used_faces_as_written(normals, angles, triangles, lookview):
    """renderer_pt3d.py:52-74, statement for statement (the eight-part comparison against the INVISIBLE vertices)."""
    B, V = angles.shape[0], normals.shape[0]
    n_t = normals[None] @ euler_angles_to_matrix_xyz(angles)
    coefs = (n_t * lookview.view(1, 1, 3)).sum(2)
    vis = torch.ones(B, V, dtype=normals.dtype)
    vis[coefs < 0] = 0
    out = []
    for b in range(B):
        idx = (vis[b] <= 0).nonzero().view(-1)
        part_num = 8
        part = int(idx.shape[0] // part_num)
        tri_vis = (~(triangles.unsqueeze(-1) == idx[:part])).any(-1)
        for j in range(1, part_num):
            sl = idx[j * part:(j + 1) * part] if j < part_num - 1 else idx[j * part:]
            tri_vis |= (~(triangles.unsqueeze(-1) == sl)).any(-1)
        out.append(triangles[(tri_vis.sum(1) > 0).nonzero().view(-1)])
    return out


def verts_normals(verts, faces):
    vn = torch.zeros_like(verts)
    vf = verts[faces]
    fn = torch.linalg.cross(vf[:, 1] - vf[:, 0], vf[:, 2] - vf[:, 0], dim=1)
    for k in range(3):
        vn = vn.index_add(0, faces[:, k], fn)
    return F.normalize(vn, eps=1e-6, dim=1)


def rasterize(verts, faces, T):
    """pix_to_face [T,T] (-1 = none) and barycentrics [T,T,3]: brute force over every (texel, face) pair."""
    j = (2 * torch.arange(T, dtype=verts.dtype) + 1) / T - 1
    px, py = j[None, :].expand(T, T), (-j)[:, None].expand(T, T)
    v = verts[faces]                                                   # [F,3,3]
    x, y, z = v[..., 0], v[..., 1], CAMERA_DISTANCE - v[..., 2]
    P = lambda t: t[None, None, :]

    def edge(ax, ay, bx, by):                                          # twice the signed area of (a, b, p), [T,T,F]
        return (P(bx) - P(ax)) * (py[..., None] - P(ay)) - (P(by) - P(ay)) * (px[..., None] - P(ax))

    area = (x[:, 1] - x[:, 0]) * (y[:, 2] - y[:, 0]) - (y[:, 1] - y[:, 0]) * (x[:, 2] - x[:, 0])
    w0 = edge(x[:, 1], y[:, 1], x[:, 2], y[:, 2]) / P(area)
    w1 = edge(x[:, 2], y[:, 2], x[:, 0], y[:, 0]) / P(area)
    w2 = edge(x[:, 0], y[:, 0], x[:, 1], y[:, 1]) / P(area)
    depth = w0 * P(z[:, 0]) + w1 * P(z[:, 1]) + w2 * P(z[:, 2])
    inside = (w0 > 0) & (w1 > 0) & (w2 > 0) & (P(area).abs() > 1e-8) & (depth >= 0)
    depth = torch.where(inside, depth, torch.full_like(depth, float("inf")))
    best = depth.min(dim=2)
    pix_to_face = torch.where(torch.isfinite(best.values), best.indices, torch.full_like(best.indices, -1))
    take = best.indices[..., None]
    bary = torch.stack([torch.gather(w, 2, take)[..., 0] for w in (w0, w1, w2)], -1)
    return pix_to_face, bary


def sample(normals, angles, triangles, imgs, template_uvs3d, face_project, lookview, texture_size, light=None):
    """-> uv_images [B,T,T,4], uv_mask [B,T,T,4] (float64; differentiable wrt imgs and face_project)."""
    lt = dict(DEFAULT_LIGHT, **(light or {}))
    dt = torch.float64
    T, B = int(texture_size), angles.shape[0]
    normals, angles, verts, lookview = (t.detach().to(dt).cpu() for t in (normals, angles, template_uvs3d, lookview))
    imgs, face_project = imgs.to(dt), face_project.to(dt)
    triangles = triangles.detach().long().cpu()
    used = used_faces_as_written(normals, angles, triangles, lookview)
    cam = torch.tensor([0.0, 0.0, CAMERA_DISTANCE], dtype=dt)
    loc = torch.tensor(lt["location"], dtype=dt)
    images, masks = [], []
    for b in range(B):
        faces = used[b]
        if faces.shape[0] == 0:
            images.append(torch.zeros(T, T, 4, dtype=dt)); masks.append(torch.zeros(T, T, 4, dtype=dt))
            continue
        p2f, bary = rasterize(verts, faces, T)
        cover = p2f >= 0
        f = faces[p2f.clamp_min(0)]                                     # [T,T,3] vertex ids
        mix = lambda attr: (attr[f] * bary[..., None]).sum(2)           # barycentric interpolation of a per-vertex attribute
        uv = mix(face_project[b])
        grid = (uv * 2.0 - 1.0)[None]
        tex = F.grid_sample(torch.flip(imgs[b:b + 1], [2]), grid, mode="bilinear", align_corners=True,
                            padding_mode="border")[0].permute(1, 2, 0)  # [T,T,C]
        n = F.normalize(mix(verts_normals(verts, faces)), eps=1e-6, dim=-1)
        pos = mix(verts)
        l = F.normalize(loc - pos, eps=1e-6, dim=-1)
        v = F.normalize(cam - pos, eps=1e-6, dim=-1)
        cos = (n * l).sum(-1)
        tint = lt["ambient"] + lt["diffuse"] * torch.relu(cos)
        refl = -l + 2 * (cos[..., None] * n)
        alpha = torch.relu((v * refl).sum(-1)) * (cos > 0)
        spec = lt["specular"] * torch.pow(alpha, lt["shininess"])
        C = tex.shape[-1]
        rgb = torch.zeros(T, T, 3, dtype=dt)
        rgb[..., :C] = tex
        col = rgb * tint[..., None] + spec[..., None]
        m = cover[..., None].to(dt)
        images.append(torch.cat([col * m, m], -1))
        masks.append(torch.cat([(tint + spec)[..., None].expand(T, T, 3) * m, m], -1))
    return torch.stack(images), torch.stack(masks)
