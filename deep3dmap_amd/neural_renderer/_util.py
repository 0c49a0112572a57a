"""Small host-side helpers shared by the neural_renderer mirror."""
import numpy as np
import torch

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
