"""Attribute-access config dicts.  The reference wraps its nested config dicts in
``easydict.EasyDict`` and feature-detects with ``hasattr`` (grl/generative_models/diffusion_model/
diffusion_model.py:56-66,88-89); easydict is an optional dependency here, so any mapping is
accepted and converted to this equivalent."""


class AttrDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(i) for i in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = __setitem__


def as_config(cfg):
    """dict / EasyDict / AttrDict -> AttrDict (recursively)."""
    return cfg if isinstance(cfg, AttrDict) else AttrDict(cfg)
