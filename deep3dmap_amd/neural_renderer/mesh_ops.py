"""vertices_to_faces (gather, with the per-vertex scatter-add as its backward) and lighting.
Reference: NR/vertices_to_faces.py:4-22, NR/lighting.py:5-57, fill_back at NR/renderer.py:86,155-156."""
import numpy as np
import torch

from .. import _lib
from ._util import deterministic, f32c, vertex_adjacency


class _GatherFaces(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vertices, faces, fill_back):
        v = f32c(vertices)
        tri = faces.to(torch.int32).contiguous()
        B, V = v.shape[:2]
        Ft = tri.shape[1]
        Fp = 2 * Ft if fill_back else Ft
        out = torch.empty(B, Fp, 3, 3, dtype=torch.float32, device=v.device)
        if tri.shape[0] not in (1, B):
            raise ValueError("faces must have batch 1 (shared topology) or the batch size of vertices")
        rc = _lib.lib().d3m_gather_faces(_lib.ptr(v), _lib.ptr(tri), tri.shape[0], _lib.ptr(out), B, V, Ft,
                                         int(bool(fill_back)), _lib.stream_ptr())
        _lib.check(rc, "d3m_gather_faces")
        ctx.save_for_backward(tri)
        ctx.dims = (B, V, Ft, bool(fill_back))
        return out

    @staticmethod
    def backward(ctx, grad_faces):
        (tri,) = ctx.saved_tensors
        B, V, Ft, fill_back = ctx.dims
        g = f32c(grad_faces)
        if deterministic() and tri.shape[0] == 1:
            # d3m_set_deterministic: the same sums gathered per vertex in a fixed order (CSR adjacency of the index tensor)
            # instead of scattered with float atomics in arrival order
            off, items = vertex_adjacency(tri, V)
            gv = torch.empty(B, V, 3, dtype=torch.float32, device=g.device)
            _lib.check(_lib.lib().d3m_vertex_gather(_lib.ptr(g), None, _lib.ptr(off), _lib.ptr(items), _lib.ptr(gv), B, V, Ft,
                                                    int(fill_back), None, _lib.stream_ptr()), "d3m_vertex_gather")
            return gv, None, None
        gv = torch.zeros(B, V, 3, dtype=torch.float32, device=g.device)
        rc = _lib.lib().d3m_scatter_face_grads(_lib.ptr(g), _lib.ptr(tri), tri.shape[0], _lib.ptr(gv), B, V, Ft,
                                               int(fill_back), _lib.stream_ptr())
        _lib.check(rc, "d3m_scatter_face_grads")
        return gv, None, None


def vertices_to_faces(vertices, faces):
    """
    :param vertices: [batch size, number of vertices, 3]
    :param faces: [batch size, number of faces, 3] (int32)
    :return: [batch size, number of faces, 3, 3]
    """
    assert (vertices.ndimension() == 3)
    assert (faces.ndimension() == 3)
    assert (vertices.shape[0] == faces.shape[0])
    assert (vertices.shape[2] == 3)
    assert (faces.shape[2] == 3)
    return _GatherFaces.apply(vertices, faces, False)


def gather_faces(vertices, faces, fill_back):
    """vertices_to_faces with the fill_back copy (reversed vertex order, NR/renderer.py:86) made on the fly;
    `faces` may have batch 1 (shared topology)."""
    return _GatherFaces.apply(vertices, faces, fill_back)


def _vec3(x):
    if torch.is_tensor(x):
        x = x.detach().cpu().numpy()
    a = np.asarray(x, dtype=np.float32).reshape(-1)
    if a.size != 3:
        raise NotImplementedError("lighting: per-batch colours / directions are not supported by the HIP path")
    return (np.ctypeslib.as_ctypes(np.ascontiguousarray(a)), a)


class _Lighting(torch.autograd.Function):
    @staticmethod
    def forward(ctx, faces, textures, ia, idr, ca, cd, direction):
        f, t = f32c(faces), f32c(textures)
        n = f.shape[0] * f.shape[1]
        ts = t.shape[2]
        out = torch.empty_like(t)
        (cca, _), (ccd, _), (cdir, _) = _vec3(ca), _vec3(cd), _vec3(direction)
        rc = _lib.lib().d3m_lighting_forward(_lib.ptr(f), _lib.ptr(t), _lib.ptr(out), float(ia), float(idr), cca, ccd,
                                             cdir, n, ts, _lib.stream_ptr())
        _lib.check(rc, "d3m_lighting_forward")
        ctx.save_for_backward(f, t)
        ctx.par = (float(ia), float(idr), ca, cd, direction)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        f, t = ctx.saved_tensors
        ia, idr, ca, cd, direction = ctx.par
        g = f32c(grad_out)
        need_f, need_t = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gt = torch.empty_like(t) if need_t else None
        gf = torch.empty_like(f) if need_f else None
        (cca, _), (ccd, _), (cdir, _) = _vec3(ca), _vec3(cd), _vec3(direction)
        rc = _lib.lib().d3m_lighting_backward(_lib.ptr(f), _lib.ptr(t), _lib.ptr(g), _lib.ptr(gt), _lib.ptr(gf), ia, idr,
                                              cca, ccd, cdir, f.shape[0] * f.shape[1], t.shape[2], _lib.stream_ptr())
        _lib.check(rc, "d3m_lighting_backward")
        return gf, gt, None, None, None, None, None


def per_batch_light(*params):
    """True when a light colour / direction is given per batch entry ([bs,3], NR/lighting.py:25-30) rather than once."""
    for x in params:
        n = x.numel() if torch.is_tensor(x) else np.asarray(x).size
        if n != 3:
            return True
    return False


def _light_rows(x, bs, device):
    t = x.to(device=device, dtype=torch.float32) if torch.is_tensor(x) else torch.as_tensor(np.asarray(x, np.float32), device=device)
    return t.reshape(-1, 3).expand(bs, 3)


def _lighting_per_batch(faces, textures, ia, idr, ca, cd, direction):
    """The same light with one colour / direction per batch entry: a device-side tensor composition (a few small
    launches; the fused kernel takes one light for the whole batch)."""
    bs = faces.shape[0]
    ca, cd, direction = (_light_rows(x, bs, faces.device) for x in (ca, cd, direction))
    normal = torch.linalg.cross(faces[:, :, 0] - faces[:, :, 1], faces[:, :, 2] - faces[:, :, 1], dim=2)
    normal = normal / normal.norm(dim=2, keepdim=True).clamp_min(1e-5)                      # F.normalize(eps=1e-5)
    cosine = (normal * direction[:, None, :]).sum(2).clamp_min(0)                            # relu(n . dir)
    light = ia * ca[:, None, :] + idr * cd[:, None, :] * cosine[:, :, None]
    return textures * light[:, :, None, None, None, :]


def lighting(faces, textures, intensity_ambient=0.5, intensity_directional=0.5, color_ambient=(1, 1, 1),
             color_directional=(1, 1, 1), direction=(0, 1, 0)):
    """Per-face ambient + directional light applied to the texture cubes (NR/lighting.py:5-57).
    faces [bs,nf,3,3] (world space), textures [bs,nf,ts,ts,ts,3]; returns the lit textures (the reference
    multiplies in place and returns the same tensor).  Colours / direction: one 3-vector, or [bs,3]."""
    if per_batch_light(color_ambient, color_directional, direction):
        return _lighting_per_batch(faces, textures, intensity_ambient, intensity_directional, color_ambient,
                                   color_directional, direction)
    return _Lighting.apply(faces, textures, intensity_ambient, intensity_directional, color_ambient,
                           color_directional, direction)
