"""Camera / viewpoint transforms of neural_renderer, each a single HIP pass with an analytic adjoint
(d3m_camera_forward / d3m_camera_backward in include/d3m_raster.h).

Reference (NR = pnpmodules/neural_renderer/neural_renderer): NR/look_at.py:6-62, NR/look.py:6-53,
NR/perspective.py:6-21, NR/projection.py:6-43, NR/get_points_from_angles.py:6-24.
"""
import ctypes
import math

import torch

from .. import _lib
from ._util import as_device_f32, const_tensor, f32c


class _CameraFunction(torch.autograd.Function):
    """vertices [Bv,V,3] -> [B,V,3].  `params` is a dict of plain values / device tensors; gradients
    flow to the vertices only (camera parameters are treated as constants)."""

    @staticmethod
    def forward(ctx, vertices, params):
        v = f32c(vertices)
        B = params["batch"]
        cam, keep = _camera_struct(params, v.device)
        out = torch.empty(B, v.shape[1], 3, dtype=torch.float32, device=v.device)
        rc = _lib.lib().d3m_camera_forward(_lib.ptr(v), v.shape[0], ctypes.byref(cam), _lib.ptr(out), B, v.shape[1],
                                           _lib.stream_ptr())
        _lib.check(rc, "d3m_camera_forward")
        ctx.save_for_backward(v)
        ctx.params, ctx.keep = params, keep
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (v,) = ctx.saved_tensors
        params = ctx.params
        g = f32c(grad_out)
        cam, keep = _camera_struct(params, v.device)
        gv = torch.empty_like(v)
        rc = _lib.lib().d3m_camera_backward(_lib.ptr(v), v.shape[0], ctypes.byref(cam), _lib.ptr(g), _lib.ptr(gv),
                                            params["batch"], v.shape[1], _lib.stream_ptr())
        _lib.check(rc, "d3m_camera_backward")
        return gv, None


def _camera_struct(p, device):
    cam = _lib.D3MCamera()
    cam.mode = p["mode"]
    cam.perspective = int(bool(p.get("perspective", False)))
    cam.tan_half_width = float(p.get("width", 1.0))
    cam.orig_size = float(p.get("orig_size", 1.0))
    keep = []
    for field, key in (("rot", "rot"), ("eye_or_t", "eye_or_t"), ("K", "K"), ("dist", "dist")):
        t = p.get(key)
        if t is not None:
            keep.append(t)
            setattr(cam, field, t.data_ptr())
            setattr(cam, field + "_batch" if field != "eye_or_t" else "eye_batch", t.shape[0])
    return cam, keep


def _learnable(*params):
    """A camera parameter that requires grad (camera optimisation, e.g. neural_renderer's example4): the fused camera
    kernels only differentiate with respect to the vertices, so these few [batch, 3]-sized quantities then go through
    the device-side torch composition below and autograd."""
    return any(torch.is_tensor(p) and p.requires_grad for p in params)


def _unit(v):
    return v / v.norm(dim=-1, keepdim=True).clamp_min(1e-5)            # F.normalize(eps=1e-5), look_at.py:47-50


def _frame_torch(eye, target, up, is_look_at):
    """Rows (x, y, z) of the camera frame [b,3,3] from differentiable inputs (look_at.py:47-53, look.py:40-46)."""
    z = _unit(target - eye) if is_look_at else _unit(target)
    x = _unit(torch.linalg.cross(up.expand_as(z), z, dim=-1))
    y = _unit(torch.linalg.cross(z, x, dim=-1))
    return torch.stack((x, y, z), dim=1)


def _view_torch(vertices, eye, frame, perspective_angle):
    """(v - eye) expressed in the camera frame, then the optional perspective division (perspective.py:15-20)."""
    out = torch.einsum('bvk,bjk->bvj', vertices - eye[:, None, :], frame)
    if perspective_angle is not None:
        width = _tan_width(perspective_angle)
        zc = out[..., 2]
        out = torch.stack((out[..., 0] / zc / width, out[..., 1] / zc / width, zc), dim=-1)
    return out


def _projection_torch(vertices, K, R, t, dist, orig_size, eps):
    """projection.py:19-43 as one differentiable device-side composition (all five inputs may require grad)."""
    cam = torch.einsum('bvk,bjk->bvj', vertices, R) + t.reshape(-1, 1, 3)
    z = cam[..., 2]
    xn, yn = cam[..., 0] / (z + eps), cam[..., 1] / (z + eps)
    k1, k2, p1, p2, k3 = (dist[:, None, i] for i in range(5))
    r2 = torch.sqrt(xn ** 2 + yn ** 2) ** 2
    radial = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
    xd = xn * radial + 2 * p1 * xn * yn + p2 * (r2 + 2 * xn ** 2)
    yd = yn * radial + p1 * (r2 + 2 * yn ** 2) + 2 * p2 * xn * yn
    pix = torch.einsum('bvk,bjk->bvj', torch.stack((xd, yd, torch.ones_like(z)), dim=-1), K)
    u, v = pix[..., 0], orig_size - pix[..., 1]
    return torch.stack((2 * (u - orig_size / 2.) / orig_size, 2 * (v - orig_size / 2.) / orig_size, z), dim=-1)


def _tan_width(angle):
    # perspective.py:15-17: the angle is rounded to f32 before tan
    return float(torch.tan(torch.tensor(angle / 180 * math.pi, dtype=torch.float32)))


def _basis(eye, at_or_dir, up, is_look_at, batch, device):
    """rows (x, y, z) of the camera frame, computed on the device (d3m_camera_basis)."""
    rot = torch.empty(batch, 3, 3, dtype=torch.float32, device=device)
    rc = _lib.lib().d3m_camera_basis(_lib.ptr(eye), eye.shape[0], _lib.ptr(at_or_dir), at_or_dir.shape[0],
                                     _lib.ptr(up), up.shape[0], int(is_look_at), _lib.ptr(rot), batch,
                                     _lib.stream_ptr())
    _lib.check(rc, "d3m_camera_basis")
    return rot


def _vec_param(x, device):
    t = as_device_f32(x, device)
    return t[None, :].contiguous() if t.dim() == 1 else t


def look_at_params(vertices, eye, at=[0, 0, 0], up=[0, 1, 0], _perspective_angle=None, defer_basis=False):
    """The fused camera kernels' parameter block of look_at(vertices, eye, at, up) -- for callers that run
    d3m_camera_forward / _backward inside a larger node (rasterize._RasterizeLit) -- or None when a camera parameter
    requires grad (those go through the torch composition in look_at).  defer_basis: the frame is NOT computed here
    (d3m_camera_basis, a launch of its own) but by the caller's d3m_lit_front, which leaves it in `rot`: the block then
    carries the vectors it is made of (`basis`)."""
    if _learnable(eye, at, up):
        return None
    device = vertices.device
    eye_t, at_t, up_t = _vec_param(eye, device), _vec_param(at, device), _vec_param(up, device)
    nb = max(eye_t.shape[0], at_t.shape[0], up_t.shape[0])
    p = dict(mode=_lib.CAMERA_LOOK_AT, batch=max(vertices.shape[0], nb), eye_or_t=eye_t,
             perspective=_perspective_angle is not None,
             width=_tan_width(_perspective_angle) if _perspective_angle is not None else 1.0)
    if defer_basis:
        p["rot"] = torch.empty(nb, 3, 3, dtype=torch.float32, device=device)
        p["basis"] = (eye_t, at_t, up_t, True)
    else:
        p["rot"] = _basis(eye_t, at_t, up_t, True, nb, device)
    return p


def basis_struct(p):
    """(D3MBasis, tensors to keep alive) of a parameter block made with defer_basis, else (None, [])"""
    b = p.get("basis")
    if b is None:
        return None, []
    eye, at, up, is_look_at = b
    return _lib.D3MBasis(eye.data_ptr(), at.data_ptr(), up.data_ptr(), eye.shape[0], at.shape[0], up.shape[0],
                         int(bool(is_look_at))), [eye, at, up]


def look_at(vertices, eye, at=[0, 0, 0], up=[0, 1, 0], _perspective_angle=None):
    """"Look at" transformation of vertices (NR/look_at.py:6-62).
    `eye`, `at`, `up`: list / tuple / ndarray / tensor of shape [3] or [batch, 3]."""
    if vertices.ndimension() != 3:
        raise ValueError('vertices Tensor should have 3 dimensions')
    params = look_at_params(vertices, eye, at, up, _perspective_angle)
    if params is None:
        device = vertices.device
        eye_t, at_t, up_t = _vec_param(eye, device), _vec_param(at, device), _vec_param(up, device)
        B = max(vertices.shape[0], eye_t.shape[0], at_t.shape[0], up_t.shape[0])   # vertices of batch 1 = one mesh seen by every camera
        return _view_torch(vertices.float().expand(B, -1, -1), eye_t.expand(B, 3),
                           _frame_torch(eye_t.expand(B, 3), at_t.expand(B, 3), up_t.expand(B, 3), True),
                           _perspective_angle)
    return _CameraFunction.apply(vertices, params)


def look(vertices, eye, direction=[0, 1, 0], up=None, _perspective_angle=None):
    """"Look" transformation of vertices (NR/look.py:6-53); `up` defaults to [0, 1, 0]."""
    if vertices.ndimension() != 3:
        raise ValueError('vertices Tensor should have 3 dimensions')
    device = vertices.device
    eye_t, dir_t = _vec_param(eye, device), _vec_param(direction, device)
    up_t = _vec_param([0, 1, 0] if up is None else up, device)
    nb = max(eye_t.shape[0], dir_t.shape[0], up_t.shape[0])
    B = max(vertices.shape[0], nb)
    if _learnable(eye, direction, up):
        return _view_torch(vertices.float().expand(B, -1, -1), eye_t.expand(B, 3),
                           _frame_torch(eye_t.expand(B, 3), dir_t.expand(B, 3), up_t.expand(B, 3), False),
                           _perspective_angle)
    return _CameraFunction.apply(vertices, look_params(vertices, eye, direction, up, _perspective_angle))


def look_params(vertices, eye, direction=[0, 1, 0], up=None, _perspective_angle=None, defer_basis=False):
    """look()'s parameter block (see look_at_params), or None when a camera parameter requires grad."""
    if _learnable(eye, direction, up):
        return None
    device = vertices.device
    eye_t, dir_t = _vec_param(eye, device), _vec_param(direction, device)
    up_t = _vec_param([0, 1, 0] if up is None else up, device)
    nb = max(eye_t.shape[0], dir_t.shape[0], up_t.shape[0])
    p = dict(mode=_lib.CAMERA_LOOK, batch=max(vertices.shape[0], nb), eye_or_t=eye_t,
             perspective=_perspective_angle is not None,
             width=_tan_width(_perspective_angle) if _perspective_angle is not None else 1.0)
    if defer_basis:
        p["rot"] = torch.empty(nb, 3, 3, dtype=torch.float32, device=device)
        p["basis"] = (eye_t, dir_t, up_t, False)
    else:
        p["rot"] = _basis(eye_t, dir_t, up_t, False, nb, device)
    return p


def perspective(vertices, angle=30.):
    """Perspective distortion x,y /= z*tan(angle) (NR/perspective.py:6-21)."""
    if vertices.ndimension() != 3:
        raise ValueError('vertices Tensor should have 3 dimensions')
    device = vertices.device
    params = dict(mode=_lib.CAMERA_LOOK_AT, batch=vertices.shape[0],
                  rot=const_tensor([[1, 0, 0], [0, 1, 0], [0, 0, 1]], device, (1, 3, 3)),
                  eye_or_t=const_tensor([0, 0, 0], device, (1, 3)), perspective=True, width=_tan_width(angle))
    return _CameraFunction.apply(vertices, params)


def projection(vertices, K, R, t, dist_coeffs, orig_size, eps=1e-9):
    """Projective transformation with lens distortion (NR/projection.py:6-43).
    K [b,3,3], R [b,3,3], t [b,1,3] (or [b,3]), dist_coeffs [b,5]; b is 1 or the batch size."""
    device = vertices.device
    tt = as_device_f32(t, device).reshape(-1, 3)
    rot, Kt, dist = (as_device_f32(R, device).reshape(-1, 3, 3), as_device_f32(K, device).reshape(-1, 3, 3),
                     as_device_f32(dist_coeffs, device).reshape(-1, 5))
    batch = max(vertices.shape[0], tt.shape[0], rot.shape[0], Kt.shape[0], dist.shape[0])
    if _learnable(K, R, t, dist_coeffs) or eps != 1e-9:      # the fused kernel has the reference's default eps built in
        return _projection_torch(vertices.float().expand(batch, -1, -1), Kt.expand(batch, 3, 3), rot.expand(batch, 3, 3),
                                 tt.expand(batch, 3), dist.expand(batch, 5), float(orig_size), eps)
    return _CameraFunction.apply(vertices, projection_params(vertices, K, R, t, dist_coeffs, orig_size))


def projection_params(vertices, K, R, t, dist_coeffs, orig_size):
    """projection()'s parameter block (default eps), or None when a camera parameter requires grad."""
    if _learnable(K, R, t, dist_coeffs):
        return None
    device = vertices.device
    tt = as_device_f32(t, device).reshape(-1, 3).contiguous()
    rot, Kt, dist = (as_device_f32(R, device).reshape(-1, 3, 3).contiguous(), as_device_f32(K, device).reshape(-1, 3, 3).contiguous(),
                     as_device_f32(dist_coeffs, device).reshape(-1, 5).contiguous())
    batch = max(vertices.shape[0], tt.shape[0], rot.shape[0], Kt.shape[0], dist.shape[0])
    return dict(mode=_lib.CAMERA_PROJECTION, batch=batch, rot=rot, eye_or_t=tt, K=Kt, dist=dist, orig_size=float(orig_size))


def get_points_from_angles(distance, elevation, azimuth, degrees=True):
    """Spherical -> cartesian eye position (NR/get_points_from_angles.py:6-24).  Host-side scalar form, or
    tensor form ([n] tensors -> [n,3])."""
    if isinstance(distance, (float, int)):
        if degrees:
            elevation, azimuth = math.radians(elevation), math.radians(azimuth)
        return (distance * math.cos(elevation) * math.sin(azimuth), distance * math.sin(elevation),
                -distance * math.cos(elevation) * math.cos(azimuth))
    if degrees:
        elevation, azimuth = math.pi / 180. * elevation, math.pi / 180. * azimuth
    return torch.stack([distance * torch.cos(elevation) * torch.sin(azimuth), distance * torch.sin(elevation),
                        -distance * torch.cos(elevation) * torch.cos(azimuth)]).transpose(1, 0)
