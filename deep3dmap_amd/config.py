"""Loads deep3dmap's python-style config files (deep3dmap/core/utils/config.py:78-602 is an mmcv clone:
the file is executed and its top-level names become the config) far enough to feed the renderer
adapters: `configs/gan2shape/*.py` and `configs/pt3d_demos/*.py` load unmodified.

Only the keys on the rasterization path are consumed (deep3dmap/core/renderer/renderer_nr.py:14-22 and
models/frameworks/imgs2mesh.py:60-62); everything else (heads, datasets, optimiser, runner) is carried
along untouched for the caller."""
import os
import types


class ConfigDict(dict):
    """dict with attribute access and mmcv's `.get(key, default)` behaviour, recursively."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name, value):
        self[name] = value


def _wrap(v):
    if isinstance(v, dict):
        return ConfigDict({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, (list, tuple)):
        return type(v)(_wrap(x) for x in v)
    return v


def _merge(base, over):
    out = dict(base)
    for k, v in over.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get("_delete_", False):
            out[k] = _merge(out[k], v)
        else:
            out[k] = {kk: vv for kk, vv in v.items() if kk != "_delete_"} if isinstance(v, dict) else v
    return out


def load_config(path):
    """Execute a config file and return its public top-level names as a ConfigDict.
    `_base_ = 'other.py'` / `_base_ = [...]` inheritance is honoured (paths relative to the file)."""
    path = os.path.abspath(path)
    with open(path) as f:
        source = f.read()
    scope = {"__file__": path}
    exec(compile(source, path, "exec"), scope)          # the reference does the same (config.py:41-57)
    cfg = {k: v for k, v in scope.items()
           if not k.startswith("__") and not isinstance(v, (types.ModuleType, types.FunctionType))}
    bases = cfg.pop("_base_", None)
    if bases:
        merged = {}
        for b in ([bases] if isinstance(bases, str) else bases):
            merged = _merge(merged, load_config(os.path.join(os.path.dirname(path), b)))
        cfg = _merge(merged, cfg)
    return _wrap(cfg)


def build_renderer(cfg):
    """The renderer adapter a framework of `cfg.model.type` constructs, with the same arguments:
    Gan2Shape -> NrRenderer(model_cfgs, image_size) (models/frameworks/gan2shape.py:99).
    imgs2mesh -> Pt3dRenderer(device, texture_size, lookview=(0,0,1)) (models/frameworks/imgs2mesh.py:62)."""
    model_cfgs = cfg["model"]["model_cfgs"]
    kind = cfg["model"].get("type")
    if kind == "Gan2Shape":
        from .core.renderer_nr import NrRenderer
        return NrRenderer(model_cfgs, model_cfgs.get("image_size", 64))
    if kind == "imgs2mesh":
        import torch
        from .core.renderer_pt3d import Pt3dRenderer
        return Pt3dRenderer("cuda", model_cfgs.get("texture_size", 256), lookview=torch.tensor([0., 0., 1.]))
    raise ValueError(f"no renderer adapter for model type {kind!r}")
