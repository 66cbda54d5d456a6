"""Configuration objects: nested dicts with attribute access.

The reference reads its YAML into an EasyDict (/root/reference/pcdet/config.py:51-80) and its modules use both
`cfg.KEY` and `cfg.get('KEY', default)`; easydict is not in this image, and the YAML files themselves are absent from
the snapshot, so configurations here are plain python dicts with the same key names, wrapped in this
class where a module wants attribute access.
"""


class Config(dict):
    """dict whose keys read as attributes, recursively (lists of dicts included)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = _wrap(v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = _wrap(v)


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, Config):
        return Config(v)
    if isinstance(v, (list, tuple)):
        return type(v)(_wrap(x) for x in v)
    return v


def cfg_from_dict(d):
    return _wrap(d)
