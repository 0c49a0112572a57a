"""
oracle/nr_ref_hip.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Device-tensor front end of oracle/_ref/libnr_ref_hip.so: the reference's own six rasterizer kernels
(pnpmodules/neural_renderer/neural_renderer/cuda/rasterize_cuda_kernel.cu:22-593, abbreviated KCU) compiled
for gfx950 from the text where it lies (oracle/Makefile, target `ref_hip`).  The functions below restate
RasterizeFunction.forward / backward (NR/rasterize.py:22-156) on torch DEVICE tensors so that full-size
scenes (BASELINE configs 2, 4, 5) can be checked against the reference without a host round trip of the
faces.  Only tests/ import this module.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATHS = {"off": os.path.join(_HERE, "_ref", "libnr_ref_hip.so"),
          "fma": os.path.join(_HERE, "_ref", "libnr_ref_hip_fma.so")}
_P, _I, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
_SIGS = {
    "refhipd_face_inverse": [_P, _P, _I, _I, _I],
    "refhipd_face_index_map": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _I],
    "refhipd_texture_sampling": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F],
    "refhipd_backward_pixel_map": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _I],
    "refhipd_backward_pixel_map_f64": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, ctypes.c_double, _I, _I],
    "refhipd_backward_textures": [_P, _P, _P, _P, _P, _I, _I, _I, _I],
    "refhipd_backward_depth_map": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I],
}
_libs = {}


def available():
    return all(os.path.exists(p) for p in _PATHS.values())


def lib(contract="off"):
    """contract='off': built with -ffp-contract=off (bit-for-bit comparisons); 'fma': the compiler's default."""
    if contract not in _libs:
        h = ctypes.CDLL(_PATHS[contract])
        for name, sig in _SIGS.items():
            fn = getattr(h, name)
            fn.argtypes, fn.restype = sig, None
        _libs[contract] = h
    return _libs[contract]


def _p(t):
    assert t.is_cuda and t.is_contiguous()
    return ctypes.c_void_p(t.data_ptr())


def forward(faces, textures, image_size, near, far, eps, background=(0, 0, 0), return_rgb=True, return_alpha=True,
            return_depth=True, contract="off"):
    """NR/rasterize.py:50-84 with the reference's initial values (:50-69), K1 -> K2 (-> K3), background, alpha.
    faces [B,F,3,3] f32 device tensor.  Returns a dict of device tensors."""
    L = lib(contract)
    torch.cuda.synchronize()
    dev = faces.device
    B, Fn = faces.shape[:2]
    S = int(image_size)
    m = {"faces": faces, "image_size": S, "eps": float(eps)}
    m["face_index_map"] = torch.full((B, S, S), -1, dtype=torch.int32, device=dev)
    m["weight_map"] = torch.zeros(B, S, S, 3, device=dev)
    m["depth_map"] = torch.full((B, S, S), float(far), device=dev)
    m["face_inv_map"] = torch.zeros(B, S, S, 3, 3, device=dev) if return_depth else torch.zeros(1, device=dev)
    m["faces_inv"] = torch.zeros_like(faces)
    torch.cuda.synchronize()
    L.refhipd_face_inverse(_p(faces), _p(m["faces_inv"]), B, Fn, S)
    L.refhipd_face_index_map(_p(faces), _p(m["faces_inv"]), _p(m["face_index_map"]), _p(m["weight_map"]),
                             _p(m["depth_map"]), _p(m["face_inv_map"]), B, Fn, S, float(near), float(far),
                             int(bool(return_depth)))
    if return_rgb:
        ts = textures.shape[2]
        m["textures"] = textures
        m["rgb_sampled"] = torch.zeros(B, S, S, 3, device=dev)
        m["sampling_index_map"] = torch.zeros(B, S, S, 8, dtype=torch.int32, device=dev)
        m["sampling_weight_map"] = torch.zeros(B, S, S, 8, device=dev)
        torch.cuda.synchronize()
        L.refhipd_texture_sampling(_p(faces), _p(textures), _p(m["face_index_map"]), _p(m["weight_map"]),
                                   _p(m["depth_map"]), _p(m["rgb_sampled"]), _p(m["sampling_index_map"]),
                                   _p(m["sampling_weight_map"]), B, Fn, S, ts, float(eps))
        bg = torch.as_tensor(background, dtype=torch.float32, device=dev)
        bg = bg[None, None, None, :] if bg.dim() == 1 else bg[:, None, None, :]
        mask = (m["face_index_map"] >= 0).float()[..., None]
        m["rgb_map"] = (m["rgb_sampled"] * mask + (1 - mask) * bg).contiguous()        # NR/rasterize.py:187-195
    if return_alpha:
        m["alpha_map"] = (m["face_index_map"] >= 0).float()                             # NR/rasterize.py:181-184
    torch.cuda.synchronize()
    return m


def backward(m, grad_rgb_map, grad_alpha_map, grad_depth_map, return_rgb, return_alpha, return_depth,
             contract="off"):
    """NR/rasterize.py:109-156: zeroed grad_faces / grad_textures, K4 (overwrite) -> K5 -> K6 (add)."""
    L = lib(contract)
    faces = m["faces"]
    dev = faces.device
    B, Fn = faces.shape[:2]
    S = m["image_size"]
    grad_faces = torch.zeros_like(faces)
    grad_textures = None
    dummy = torch.zeros(1, device=dev)
    torch.cuda.synchronize()
    if return_rgb or return_alpha:
        L.refhipd_backward_pixel_map(
            _p(faces), _p(m["face_index_map"]), _p(m["rgb_map"] if return_rgb else dummy),
            _p(m["alpha_map"] if return_alpha else dummy), _p(grad_rgb_map if return_rgb else dummy),
            _p(grad_alpha_map if return_alpha else dummy), _p(grad_faces), B, Fn, S, m["eps"], int(bool(return_rgb)),
            int(bool(return_alpha)))
    if return_rgb:
        grad_textures = torch.zeros_like(m["textures"])
        L.refhipd_backward_textures(_p(m["face_index_map"]), _p(m["sampling_weight_map"]), _p(m["sampling_index_map"]),
                                    _p(grad_rgb_map), _p(grad_textures), B, Fn, S, m["textures"].shape[2])
    if return_depth:
        L.refhipd_backward_depth_map(_p(faces), _p(m["depth_map"]), _p(m["face_index_map"]), _p(m["face_inv_map"]),
                                     _p(m["weight_map"]), _p(grad_depth_map), _p(grad_faces), B, Fn, S)
    torch.cuda.synchronize()
    return grad_faces, grad_textures


def backward_pixel_map_f64(m, grad_rgb_map, grad_alpha_map, return_rgb=True, return_alpha=True):
    """K4 alone through the reference kernel's DOUBLE instantiation (KCU:245-503 with scalar_t = double; the reference
    dispatches on the faces' type, KCU:614) on the same inputs widened to f64: grad_faces [B,F,3,3] f64 -- the value both
    f32 evaluations (the reference's own and the product's) approximate."""
    L = lib("off")
    faces = m["faces"]
    dev = faces.device
    B, Fn = faces.shape[:2]
    S = m["image_size"]
    dummy = torch.zeros(1, dtype=torch.float64, device=dev)
    wide = lambda t, on: t.double().contiguous() if on else dummy
    f64 = faces.double().contiguous()
    rgb, alpha = wide(m.get("rgb_map"), return_rgb), wide(m.get("alpha_map"), return_alpha)
    g_rgb, g_alpha = wide(grad_rgb_map, return_rgb), wide(grad_alpha_map, return_alpha)
    grad_faces = torch.zeros_like(f64)
    torch.cuda.synchronize()
    # (eps as the f32 value the float instantiation receives, widened)
    L.refhipd_backward_pixel_map_f64(_p(f64), _p(m["face_index_map"]), _p(rgb), _p(alpha), _p(g_rgb), _p(g_alpha),
                                     _p(grad_faces), B, Fn, S, float(torch.tensor(m["eps"], dtype=torch.float32)),
                                     int(bool(return_rgb)), int(bool(return_alpha)))
    torch.cuda.synchronize()
    return grad_faces
