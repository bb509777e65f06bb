"""Minimal stand-in for ``mmcv.Config`` (not installed here): loads the reference's Python-file configs
unchanged and exposes attribute + dict access with ``copy()/get()/pop()`` -- everything the builders of the
hot path touch ("Config / flags")."""
import runpy
import types


class ConfigDict(dict):
    """dict with attribute access; nested dicts are converted recursively; ``copy()`` stays a ConfigDict."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, ConfigDict):
            return v
        if isinstance(v, dict):
            return ConfigDict(v)
        if isinstance(v, (list, tuple)):
            return type(v)(ConfigDict._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, ConfigDict._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError("'ConfigDict' object has no attribute '%s'" % k)

    def __setattr__(self, k, v):
        self[k] = v

    def __delattr__(self, k):
        del self[k]

    def copy(self):
        return ConfigDict(self)

    def update(self, *args, **kwargs):
        for k, v in dict(*args, **kwargs).items():
            self[k] = v


class Config(ConfigDict):
    """``Config.fromfile('configs/PSMNet/scene_flow.py')`` or ``Config(dict(model=...))``."""

    @staticmethod
    def fromfile(filename):
        ns = runpy.run_path(filename)
        keep = {k: v for k, v in ns.items()
                if not k.startswith("_") and not isinstance(v, (types.ModuleType, types.FunctionType, type))}
        return Config(keep)
