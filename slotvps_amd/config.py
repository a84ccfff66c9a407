"""Minimal python-file config loader with attribute-dict semantics (the subset of mmcv.Config the
reference's configs use: ``Config.fromfile(path)`` then ``cfg.model``, ``cfg.test_cfg``, nested
attribute access such as ``pos_config.hidden_dim``, position_encoding.py:337-338)."""
import os
import runpy


class ConfigDict(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name, value):
        self[name] = value


def _wrap(x):
    if isinstance(x, dict):
        return ConfigDict({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    if isinstance(x, tuple):
        return tuple(_wrap(v) for v in x)
    return x


class Config:
    def __init__(self, cfg_dict, filename=None):
        object.__setattr__(self, "_cfg_dict", _wrap(cfg_dict))
        object.__setattr__(self, "filename", filename)

    @staticmethod
    def fromfile(filename):
        filename = os.path.abspath(os.path.expanduser(filename))
        if not os.path.isfile(filename):
            raise FileNotFoundError(filename)
        ns = runpy.run_path(filename)
        cfg = {k: v for k, v in ns.items() if not k.startswith("__") and not callable(v)
               and not isinstance(v, type(os))}
        return Config(cfg, filename)

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def get(self, key, default=None):
        return self._cfg_dict.get(key, default)

    def __contains__(self, key):
        return key in self._cfg_dict
