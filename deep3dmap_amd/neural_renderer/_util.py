"""Small host-side helpers shared by the neural_renderer mirror."""
import numpy as np
import torch

from .. import _lib

_const_cache = {}


def const_tensor(values, device, shape=None):
    """Device tensor for a small host constant (list / tuple / ndarray / float), uploaded once per
    (value, device): `torch.tensor(list, device=...)` is a synchronous copy, and the reference pays it
    on every call (look_at.py:16-37, rasterize.py:189)."""
    arr = np.asarray(values, dtype=np.float32)
    key = (arr.tobytes(), arr.shape, str(device), shape)
    t = _const_cache.get(key)
    if t is None:
        t = torch.from_numpy(arr.copy()).to(device)
        if shape is not None:
            t = t.reshape(shape)
        t = t.contiguous()
        if len(_const_cache) > 4096:
            _const_cache.clear()
        _const_cache[key] = t
    return t


def as_device_f32(x, device):
    """list / tuple / ndarray / tensor -> contiguous f32 tensor on `device`."""
    if torch.is_tensor(x):
        return x.to(device=device, dtype=torch.float32).contiguous()
    return const_tensor(x, device)


def f32c(t):
    """contiguous f32 view/copy of a tensor (the C ABI is f32-only)."""
    return t.to(torch.float32).contiguous()


_adjacency_cache = {}


def deterministic():
    """d3m_set_deterministic / D3M_DETERMINISTIC=1: see _RasterizeLit._backward_halves, "DETERMINISTIC"."""
    return _lib.lib().d3m_get_deterministic() == 1


def vertex_adjacency(tri, num_vertices):
    """CSR adjacency of ONE index tensor tri [1,F,3] (int32, device): (offsets [V+1], items [3F]) with item = 3 f + c for
    "corner c of triangle f", a vertex's items in ascending order -- what d3m_vertex_gather walks.  Built with torch
    operators once per (tensor, version) and cached (a handful of topologies per process); build it OUTSIDE a stream
    capture (MultiViewFit's warm-up steps do)."""
    key = (tri.data_ptr(), tri._version, tuple(tri.shape), int(num_vertices), str(tri.device))
    hit = _adjacency_cache.get(key)
    if hit is None:
        flat = tri.reshape(-1).long()
        items = torch.argsort(flat, stable=True).to(torch.int32).contiguous()
        offsets = torch.zeros(int(num_vertices) + 1, dtype=torch.int32, device=tri.device)
        offsets[1:] = torch.cumsum(torch.bincount(flat, minlength=int(num_vertices)), 0).to(torch.int32)
        if len(_adjacency_cache) >= 16:
            _adjacency_cache.clear()
        hit = _adjacency_cache[key] = (offsets.contiguous(), items, tri)       # (the tensor itself: its address stays taken)
    return hit[0], hit[1]
