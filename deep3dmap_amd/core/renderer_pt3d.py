"""Pt3dRenderer: deep3dmap's UV-space unwrapping (deep3dmap/core/renderer/renderer_pt3d.py:32-98; caller
models/frameworks/imgs2mesh.py:100-124), same constructor and `sample` signature, on the native rasterizer.

The reference delegates the arithmetic to pytorch3d (pinned 0.6.1, third party, not vendored): euler_angles_to_matrix,
Rotate.transform_normals, look_at_view_transform + OpenGLOrthographicCameras, MeshRasterizer (hard, one face per pixel),
TexturesUV, SoftPhongShader with a point light.  What those compute is restated here from the library's documented
behaviour and built on this repository's operators; there is no reference vector to pin it to, so the
tests are analytic (tests/test_gpu_pt3d.py).

  sample(normals [V,3], angles [B,3], triangles [F,3], imgs [B,C,H,W], template_uvs3d [V,3], face_project [B,V,2])
      -> uv_images [B,T,T,4], uv_mask [B,T,T,4]                                   (T = texture_size, row 0 = top)

1. per-vertex visibility: normals rotated by the XYZ Euler angles, visible iff n' . lookview >= 0 (renderer_pt3d.py:52-57);
2. the triangle filter of renderer_pt3d.py:59-74 AS WRITTEN: it compares every triangle corner with the indices of the
   INVISIBLE vertices and keeps a triangle if any corner differs from any of them -- which keeps every triangle as soon
   as one vertex of the mesh is invisible, and none when all are visible (the parts of the index list are then empty);
3. the mesh template_uvs3d / kept triangles seen by an orthographic camera from (0, 0, 2.7) (NDC x, y = world x, y),
   rasterized hard with both windings (pytorch3d does not cull): d3m_forward_face_index_map on the fill_back faces;
4. per covered pixel (d3m_uv_unwrap): bilinear lookup of imgs at the barycentric mix of face_project (TexturesUV:
   align_corners, border padding, v measured upwards), shaded by pytorch3d's Phong model, background 0.  WHICH light:
   the reference builds PointLights at (0, 0, 10) with ambient 0 / diffuse 1 / specular 0 and matching Materials
   (renderer_pt3d.py:42-43) but never hands them to SoftPhongShader(device, cameras, blend_params) or to renderer(mesh)
   (:87-93), so pytorch3d 0.6.1 shades with its DEFAULTS: PointLights at (0, 1, 0), ambient 0.5, diffuse 0.3, specular
   0.2; Materials 1 / 1 / 1, shininess 64 -- colour = (0.5 + 0.3 relu(n.l)) texel + 0.2 relu(v.r)^64 [n.l > 0], view
   vector towards the camera at (0, 0, 2.7).  That is what runs, so that is what is reproduced (the class attributes
   below hold the values; set them to the reference's unused ones for the light its author meant).  The alpha channel is
   the hard coverage (pytorch3d's softmax blend gives a sigmoid of the distance to the face's edges; the caller reads
   rgb only, imgs2mesh.py:121-122).
"""
import ctypes

import torch

from .. import _lib
from .. import neural_renderer as nr
from ..neural_renderer import rasterize_ops as ops
from ..neural_renderer._util import f32c
from ..neural_renderer.mesh_ops import gather_faces
from .losses import silhouette_loss


def euler_xyz_to_matrix(angles):
    """pytorch3d.transforms.euler_angles_to_matrix(angles, "XYZ") = Rx(a0) Ry(a1) Rz(a2)."""
    c, s = torch.cos(angles), torch.sin(angles)
    one, zero = torch.ones_like(c[:, 0]), torch.zeros_like(c[:, 0])
    rx = torch.stack([one, zero, zero, zero, c[:, 0], -s[:, 0], zero, s[:, 0], c[:, 0]], 1).reshape(-1, 3, 3)
    ry = torch.stack([c[:, 1], zero, s[:, 1], zero, one, zero, -s[:, 1], zero, c[:, 1]], 1).reshape(-1, 3, 3)
    rz = torch.stack([c[:, 2], -s[:, 2], zero, s[:, 2], c[:, 2], zero, zero, zero, one], 1).reshape(-1, 3, 3)
    return rx.matmul(ry).matmul(rz)


def vertex_normals(verts, triangles):
    """pytorch3d Meshes.verts_normals_packed: area-weighted sum of the face normals at every vertex, normalised."""
    tri = triangles.long()
    v0, v1, v2 = verts[tri[:, 0]], verts[tri[:, 1]], verts[tri[:, 2]]
    fn = torch.linalg.cross(v1 - v0, v2 - v0, dim=1)             # length = twice the area
    vn = torch.zeros_like(verts)
    for k in range(3):
        vn.index_add_(0, tri[:, k], fn)
    return torch.nn.functional.normalize(vn, eps=1e-6, dim=1)


class _UvUnwrap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, imgs, uvs, fi, wm, tri, verts, vnormals, used, light, T):
        im, uv = f32c(imgs), f32c(uvs)
        B, C, H, W = im.shape
        out_img = torch.empty(B, T, T, 4, dtype=torch.float32, device=im.device)
        out_mask = torch.empty(B, T, T, 4, dtype=torch.float32, device=im.device)
        lt = (ctypes.c_float * 10)(*light)
        _lib.check(_lib.lib().d3m_uv_unwrap(_lib.ptr(fi), _lib.ptr(wm), _lib.ptr(tri), _lib.ptr(verts), _lib.ptr(vnormals),
                                            _lib.ptr(uv), _lib.ptr(im), _lib.ptr(used), lt, _lib.ptr(out_img),
                                            _lib.ptr(out_mask), B, fi.shape[0], T, tri.shape[0], verts.shape[0], C, H, W,
                                            _lib.stream_ptr()), "d3m_uv_unwrap")
        ctx.save_for_backward(im, uv, fi, wm, tri, verts, vnormals, used)
        ctx.light, ctx.T = tuple(light), T
        ctx.mark_non_differentiable(out_mask)
        return out_img, out_mask

    @staticmethod
    def backward(ctx, g_img, g_mask):
        im, uv, fi, wm, tri, verts, vnormals, used = ctx.saved_tensors
        B, C, H, W = im.shape
        g_im = torch.zeros_like(im) if ctx.needs_input_grad[0] else None
        g_uv = torch.zeros_like(uv) if ctx.needs_input_grad[1] else None
        lt = (ctypes.c_float * 10)(*ctx.light)
        _lib.check(_lib.lib().d3m_uv_unwrap_backward(
            _lib.ptr(fi), _lib.ptr(wm), _lib.ptr(tri), _lib.ptr(verts), _lib.ptr(vnormals), _lib.ptr(uv), _lib.ptr(im),
            _lib.ptr(used), lt, _lib.ptr(f32c(g_img)), _lib.ptr(g_im), _lib.ptr(g_uv), B, fi.shape[0], ctx.T, tri.shape[0],
            verts.shape[0], C, H, W, _lib.stream_ptr()), "d3m_uv_unwrap_backward")
        return (g_im, g_uv) + (None,) * 8


class Pt3dRenderer():
    CAMERA_DISTANCE = 2.7                    # look_at_view_transform(2.7, 0, 0), renderer_pt3d.py:78
    # what SoftPhongShader falls back to when it is given no lights / materials (pytorch3d 0.6.1 PointLights / Materials
    # defaults; the reference's own PointLights((0,0,10), 0/1/0) and Materials are constructed but never used)
    LIGHT_LOCATION = (0.0, 1.0, 0.0)
    LIGHT_AMBIENT, LIGHT_DIFFUSE, LIGHT_SPECULAR, SHININESS = 0.5, 0.3, 0.2, 64.0

    def __init__(self, device, texture_size, lookview):
        self.device = device
        self.texture_size = texture_size
        self.lookview = lookview.view(1, 3).to(torch.float32)
        self._coverage = None                # the UV layout is a constant of the model: its raster is computed once

    def _light(self):
        """d3m_uv_unwrap's light block: location, camera centre, ambient, diffuse, specular, shininess"""
        return tuple(self.LIGHT_LOCATION) + (0.0, 0.0, self.CAMERA_DISTANCE, self.LIGHT_AMBIENT, self.LIGHT_DIFFUSE,
                                             self.LIGHT_SPECULAR, self.SHININESS)

    def _uv_coverage(self, template_uvs3d, triangles):
        """face_index_map / weight_map [1,T,T(,3)] of the template in UV space (orthographic: NDC xy = world xy,
        depth = camera distance - z), both windings; cached while the SAME tensors (kept alive here, compared by identity
        and version) are passed again."""
        c = self._coverage
        if (c is not None and c[0] is template_uvs3d and c[1] is triangles and
                c[2] == (template_uvs3d._version, triangles._version)):
            return c[3]
        T = self.texture_size
        v = f32c(template_uvs3d)
        screen = torch.stack((v[:, 0], v[:, 1], self.CAMERA_DISTANCE - v[:, 2]), 1)[None]
        faces = gather_faces(screen, triangles.to(torch.int32)[None], True)
        fi = torch.empty(1, T, T, dtype=torch.int32, device=v.device)
        wm = torch.empty(1, T, T, 3, dtype=torch.float32, device=v.device)
        dm = torch.empty(1, T, T, dtype=torch.float32, device=v.device)
        dummy = torch.zeros(1, dtype=torch.float32, device=v.device)
        ops.forward_face_index_map(faces, fi, wm, dm, dummy, dummy, T, 1e-4, 100.0, False, True, False)
        hit = (fi, wm, vertex_normals(v, triangles))
        self._coverage = (template_uvs3d, triangles, (template_uvs3d._version, triangles._version), hit)
        return hit

    def sample(self, normals, angles, triangles, imgs, template_uvs3d, face_project):
        batchsize = angles.shape[0]
        T = self.texture_size
        dev = imgs.device
        # 1. per-vertex visibility (renderer_pt3d.py:52-57): Rotate(RR).transform_normals(n) = n @ RR for a rotation
        n_rot = f32c(normals).matmul(euler_xyz_to_matrix(f32c(angles)))                       # [B,V,3]
        invisible = (n_rot * self.lookview.to(dev)).sum(2) < 0
        # 2. the triangle filter as written (see the module docstring): every triangle, unless no vertex is invisible
        used = invisible.any(dim=1).to(torch.int32)
        # 3. coverage of the UV layout, shared by the batch
        fi, wm, vnormals = self._uv_coverage(template_uvs3d, triangles)
        # 4. texture lookup + Phong shading (the one coverage map is shared by the batch)
        return _UvUnwrap.apply(imgs, face_project, fi, wm, triangles.to(torch.int32).contiguous(), f32c(template_uvs3d),
                               vnormals, used, self._light(), T)

    # ---- BASELINE.json config 1 ("pt3d_demos: icosphere, 2 views @64x64, silhouette fit"): demo plumbing ------------
    def silhouettes(self, vertices, faces, azimuths, elevation=0.0, image_size=None):
        """vertices [V,3], faces [F,3] -> silhouettes [n_views, s, s] seen from `azimuths` (degrees) at the camera distance."""
        r = nr.Renderer(image_size=image_size or self.texture_size, camera_mode="look_at", anti_aliasing=True)
        r.eye = torch.tensor([nr.get_points_from_angles(self.CAMERA_DISTANCE, elevation, float(a)) for a in azimuths],
                             dtype=torch.float32, device=vertices.device)
        n = len(azimuths)
        return r(vertices[None].expand(n, -1, -1), faces[None].expand(n, -1, -1), mode="silhouettes")

    def fit(self, vertices, faces, target, azimuths, steps=50, lr=0.01, image_size=None):
        """Silhouette fit with Adam (pnpmodules/neural_renderer/examples/example2.py:43-47,62): the loss history."""
        opt = torch.optim.Adam([vertices], lr=lr)
        history = []
        for _ in range(steps):
            opt.zero_grad()
            loss = silhouette_loss(self.silhouettes(vertices, faces, azimuths, image_size=image_size), target)
            loss.backward()
            opt.step()
            history.append(float(loss.detach()))
        return history
