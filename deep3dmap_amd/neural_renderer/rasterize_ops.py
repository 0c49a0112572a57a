"""Stand-in for the pybind module `neural_renderer.cuda.rasterize`
(pnpmodules/neural_renderer/neural_renderer/cuda/rasterize_cuda.cpp:193-199): the same five functions,
same argument order, same ownership rule (caller allocates and pre-fills, callee mutates in place and
returns the same tensors), same errors (non-CUDA / non-contiguous -> RuntimeError).  Each call is one
entry point of libd3m_raster.so; kernels run on torch's current stream."""
import ctypes

import torch

from .. import _lib

_workspaces = {}


def _workspace(key, nbytes, device):
    """Scratch buffers are cached per (kind, device, stream) and only ever grow: operators that run concurrently on
    different streams (view groups, the side branch of the lit backward) must not share scratch."""
    k = (key, device, torch.cuda.current_stream(device).cuda_stream)
    buf = _workspaces.get(k)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _workspaces[k] = buf
    return buf


def visibility(face_index_map, num_faces):
    """The flags + compacted list of the faces that own a pixel (d3m_visibility), shared by the backward operators
    of one forward result.  Returns the (cached, reused) blob."""
    L = _lib.lib()
    B, S = face_index_map.shape[0], face_index_map.shape[1]
    blob = _workspace("visibility", L.d3m_visibility_bytes(B, num_faces), face_index_map.device)
    _lib.check(L.d3m_visibility(_lib.ptr(face_index_map), _lib.ptr(blob), blob.numel(), B, num_faces, S,
                                _lib.stream_ptr()), "d3m_visibility")
    return blob


def edge_plan(faces, face_index_map, visibility_blob, image_size, out=None):
    """Where the visible faces' edges cross the pixel grid, grouped by image line (d3m_edge_plan): geometry only, so it
    can be built right after the forward pass and handed to backward_pixel_map.  Returns the blob."""
    L = _lib.lib()
    B, F = faces.shape[:2]
    blob = out if out is not None else torch.empty(int(L.d3m_edge_plan_bytes(B, F, int(image_size))), dtype=torch.uint8,
                                                   device=faces.device)
    _lib.check(L.d3m_edge_plan(_lib.ptr(faces), _lib.ptr(face_index_map), _lib.ptr(visibility_blob), _lib.ptr(blob), blob.numel(), B, F,
                               int(image_size), 0, _lib.stream_ptr()), "d3m_edge_plan")
    return blob


def _opt(t):
    """The reference passes 1-element dummies for disabled outputs (rasterize.py:46,59-69)."""
    return None if (t is None or t.numel() <= 1) else t


def forward_face_index_map(faces, face_index_map, weight_map, depth_map, face_inv_map, faces_inv, image_size,
                           near, far, return_rgb, return_alpha, return_depth):
    _lib.require_device(faces, face_index_map, weight_map, depth_map, face_inv_map, faces_inv,
                        names=["faces", "face_index_map", "weight_map", "depth_map", "face_inv_map", "faces_inv"])
    L = _lib.lib()
    B, F = faces.shape[:2]
    ws = _workspace("fwd", L.d3m_forward_workspace_bytes(B, F, image_size), faces.device)
    rc = L.d3m_forward_face_index_map(_lib.ptr(faces), _lib.ptr(face_index_map), _lib.ptr(weight_map),
                                      _lib.ptr(depth_map), _lib.ptr(_opt(face_inv_map) if return_depth else None),
                                      _lib.ptr(_opt(faces_inv)), B, F, int(image_size), float(near), float(far),
                                      int(bool(return_rgb)), int(bool(return_alpha)), int(bool(return_depth)),
                                      _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
    _lib.check(rc, "forward_face_index_map")
    return [face_index_map, weight_map, depth_map, face_inv_map]


def forward_texture_sampling(faces, textures, face_index_map, weight_map, depth_map, rgb_map, sampling_index_map,
                             sampling_weight_map, image_size, eps):
    _lib.require_device(faces, textures, face_index_map, weight_map, depth_map, rgb_map, sampling_index_map,
                        sampling_weight_map,
                        names=["faces", "textures", "face_index_map", "weight_map", "depth_map", "rgb_map",
                               "sampling_index_map", "sampling_weight_map"])
    B, F = faces.shape[:2]
    rc = _lib.lib().d3m_forward_texture_sampling(
        _lib.ptr(faces), _lib.ptr(textures), _lib.ptr(face_index_map), _lib.ptr(weight_map), _lib.ptr(depth_map),
        _lib.ptr(rgb_map), _lib.ptr(_opt(sampling_index_map)), _lib.ptr(_opt(sampling_weight_map)), B, F,
        int(image_size), int(textures.shape[2]), float(eps), _lib.stream_ptr())
    _lib.check(rc, "forward_texture_sampling")
    return [rgb_map, sampling_index_map, sampling_weight_map]


def backward_pixel_map(faces, face_index_map, rgb_map, alpha_map, grad_rgb_map, grad_alpha_map, grad_faces,
                       image_size, eps, return_rgb, return_alpha, vertex_target=None, visibility=None, unscaled=None,
                       edge_plan=None):
    """`vertex_target` (a _lib.D3MVertexTarget, not part of the reference signature) sends the face gradients
    straight into the gradient of the vertices the faces were gathered from; grad_faces may then be None.
    `unscaled` (a _lib.D3MFitTargets): the gradient maps are the unscaled ones a fused fit objective left.
    `edge_plan`: the blob edge_plan() built for the same faces / visibility (the op builds its own otherwise)."""
    _lib.require_device(faces, face_index_map, rgb_map, alpha_map, grad_rgb_map, grad_alpha_map, grad_faces,
                        names=["faces", "face_index_map", "rgb_map", "alpha_map", "grad_rgb_map", "grad_alpha_map",
                               "grad_faces"])
    L = _lib.lib()
    B, F = faces.shape[:2]
    ws = _workspace("edge", L.d3m_backward_pixel_map_workspace_bytes(B, F, image_size), faces.device)
    rc = L.d3m_backward_pixel_map(
        _lib.ptr(faces), _lib.ptr(face_index_map), _lib.ptr(rgb_map if return_rgb else None),
        _lib.ptr(alpha_map if return_alpha else None), _lib.ptr(grad_rgb_map if return_rgb else None),
        _lib.ptr(grad_alpha_map if return_alpha else None), _lib.ptr(grad_faces), B, F, int(image_size), float(eps),
        int(bool(return_rgb)), int(bool(return_alpha)), _lib.ptr(ws), ws.numel(),
        ctypes.byref(vertex_target) if vertex_target is not None else None, _lib.ptr(visibility), _lib.ptr(edge_plan),
        edge_plan.numel() if edge_plan is not None else 0,
        ctypes.byref(unscaled) if unscaled is not None else None, _lib.stream_ptr())
    _lib.check(rc, "backward_pixel_map")
    return grad_faces


def backward_textures(face_index_map, sampling_weight_map, sampling_index_map, grad_rgb_map, grad_textures,
                      num_faces, faces=None):
    """`faces` is optional and not part of the reference signature: with it the texel sums are gathered per
    visible face instead of scattered with float atomics."""
    _lib.require_device(face_index_map, sampling_weight_map, sampling_index_map, grad_rgb_map, grad_textures, faces,
                        names=["face_index_map", "sampling_weight_map", "sampling_index_map", "grad_rgb_map",
                               "grad_textures", "faces"])
    L = _lib.lib()
    B, S = face_index_map.shape[:2]
    ws = _workspace("faces", L.d3m_backward_faces_workspace_bytes(B, int(num_faces)), face_index_map.device) \
        if faces is not None else None
    rc = L.d3m_backward_textures(
        _lib.ptr(faces), _lib.ptr(face_index_map), _lib.ptr(sampling_weight_map), _lib.ptr(sampling_index_map),
        _lib.ptr(grad_rgb_map), _lib.ptr(grad_textures), B, int(num_faces), S, int(grad_textures.shape[2]),
        _lib.ptr(ws), ws.numel() if ws is not None else 0, _lib.stream_ptr())
    _lib.check(rc, "backward_textures")
    return grad_textures


def backward_depth_map(faces, depth_map, face_index_map, face_inv_map, weight_map, grad_depth_map, grad_faces,
                       image_size):
    _lib.require_device(faces, depth_map, face_index_map, face_inv_map, weight_map, grad_depth_map, grad_faces,
                        names=["faces", "depth_map", "face_index_map", "face_inv_map", "weight_map",
                               "grad_depth_map", "grad_faces"])
    L = _lib.lib()
    B, F = faces.shape[:2]
    ws = _workspace("faces", L.d3m_backward_faces_workspace_bytes(B, F), faces.device)
    rc = L.d3m_backward_depth_map(
        _lib.ptr(faces), _lib.ptr(depth_map), _lib.ptr(face_index_map), _lib.ptr(_opt(face_inv_map)),
        _lib.ptr(weight_map), _lib.ptr(grad_depth_map), _lib.ptr(grad_faces), B, F, int(image_size),
        _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
    _lib.check(rc, "backward_depth_map")
    return grad_faces
