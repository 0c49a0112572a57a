"""Stand-in for the pybind module `neural_renderer.cuda.rasterize`
(pnpmodules/neural_renderer/neural_renderer/cuda/rasterize_cuda.cpp:193-199): the same five functions,
same argument order, same ownership rule (caller allocates and pre-fills, callee mutates in place and
returns the same tensors), same errors (non-CUDA / non-contiguous -> RuntimeError).  Each call is one
entry point of libd3m_raster.so; kernels run on torch's current stream."""
import ctypes

import torch

from .. import _lib

class StreamKeyedCache:
    """Per-stream objects (scratch buffers, side streams) keyed by (kind, ..., raw stream handle), BOUNDED: at most
    `max_per_kind` entries per kind, least recently used first out.  A handle says nothing about the life of the stream
    behind it -- torch.cuda.current_stream() returns a fresh wrapper object on every call, so there is nothing to hang a
    weak reference on, and torch hands out its pooled streams' handles again and again -- so entries are not tied to stream
    destruction; what is bounded is what a process that keeps switching streams can pile up (up to round 5: one scratch
    buffer of up to gigabytes per (kind, stream) ever seen, for the life of the process).  Eviction only drops the cache's
    reference: a buffer was allocated under the stream it is used on, so the caching allocator reuses its memory in that
    stream's order.  Pure host logic (tests/test_host_logic.py)."""

    def __init__(self, max_per_kind=8):
        self.max_per_kind = int(max_per_kind)
        self._items = {}            # key -> value, in order of last use (dicts keep insertion order)

    def get(self, key):
        v = self._items.pop(key, None)
        if v is not None:
            self._items[key] = v    # most recently used last
        return v

    def put(self, key, value):
        self._items.pop(key, None)
        self._items[key] = value
        same = [k for k in self._items if k[0] == key[0]]
        for k in same[:max(0, len(same) - self.max_per_kind)]:
            del self._items[k]

    def clear(self):
        self._items.clear()

    def __len__(self):
        return len(self._items)


_workspaces = StreamKeyedCache(max_per_kind=8)


# Scratch handed out INSIDE a stream capture is baked into the captured graph as raw addresses: such a buffer must outlive
# the graph whatever the cache does with its entry.  It is remembered here until the owner of the capture claims it
# (graph.CapturedStep.capture -> take_captured_refs(): the buffers then live exactly as long as the step's graphs); a
# capture nobody claims for keeps them for the life of the process, which is what every buffer did up to round 5.
_captured_refs = {}


def take_captured_refs():
    """The scratch buffers handed out under a capture since the last call (and forget them here)."""
    refs = list(_captured_refs.values())
    _captured_refs.clear()
    return refs


def release_workspaces():
    """Drop every cached scratch buffer (they are re-allocated on demand; buffers a captured graph uses stay alive
    through _captured_refs / their CapturedStep)."""
    _workspaces.clear()


def _workspace(key, nbytes, device):
    """Scratch buffers are cached per (kind, device, stream) and only ever grow: operators that run concurrently on
    different streams (view groups, the side branch of the lit backward) must not share scratch.  At most eight streams
    per kind and the least recently used goes (StreamKeyedCache)."""
    k = (key, str(device), torch.cuda.current_stream(device).cuda_stream)
    buf = _workspaces.get(k)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _workspaces.put(k, buf)
    if torch.cuda.is_current_stream_capturing():
        _captured_refs[id(buf)] = buf
    return buf


def visibility(face_index_map, num_faces, out=None):
    """The flags + compacted list of the faces that own a pixel (d3m_visibility), shared by the backward operators
    of one forward result.  Returns the blob: `out` if given (at least d3m_visibility_bytes), else a FRESH buffer per
    call -- up to round 5 this was one cached blob per stream, so that two live forward results on a stream aliased
    (tests/test_gpu_ops.py::test_visibility_blobs_do_not_alias)."""
    L = _lib.lib()
    B, S = face_index_map.shape[0], face_index_map.shape[1]
    blob = out if out is not None else torch.empty(int(L.d3m_visibility_bytes(B, num_faces)), dtype=torch.uint8,
                                                   device=face_index_map.device)
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
