"""Name -> class registries and config-driven builders with the reference's semantics
(mmdet/utils/registry.py:6-76, mmdet/models/registry.py:3-8, mmdet/models/builder.py:7-33):
bare-decorator ``@X.register_module``, duplicate names raise KeyError, ``build_from_cfg`` pops
``type`` (a registered name or a class), fills ``default_args`` with setdefault and calls the class.
"""
import inspect

from torch import nn


class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    def __repr__(self):
        return f"{self.__class__.__name__}(name={self._name}, items={list(self._module_dict)})"

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key, None)

    def register_module(self, cls):
        if not inspect.isclass(cls):
            raise TypeError(f"module must be a class, but got {type(cls)}")
        name = cls.__name__
        if name in self._module_dict:
            raise KeyError(f"{name} is already registered in {self._name}")
        self._module_dict[name] = cls
        return cls


def build_from_cfg(cfg, registry, default_args=None):
    assert isinstance(cfg, dict) and "type" in cfg
    assert isinstance(default_args, dict) or default_args is None
    args = dict(cfg)
    obj_type = args.pop("type")
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError(f"{obj_type} is not in the {registry.name} registry")
    elif inspect.isclass(obj_type):
        obj_cls = obj_type
    else:
        raise TypeError(f"type must be a str or valid type, but got {type(obj_type)}")
    if default_args is not None:
        for k, v in default_args.items():
            args.setdefault(k, v)
    return obj_cls(**args)


BACKBONES = Registry("backbone")
NECKS = Registry("neck")
PANOPTIC = Registry("panoptic")
HEADS = Registry("head")
LOSSES = Registry("loss")
DETECTORS = Registry("detector")


def _register_all():
    """The registering modules are imported on first use, not with the package (they pull in torch.nn
    model code that the kernel-level entry points do not need)."""
    from . import backbones, detector, swin  # noqa: F401


def build(cfg, registry, default_args=None):
    _register_all()
    if isinstance(cfg, list):
        return nn.Sequential(*[build_from_cfg(c, registry, default_args) for c in cfg])
    return build_from_cfg(cfg, registry, default_args)


def build_backbone(cfg):
    return build(cfg, BACKBONES)


def build_neck(cfg):
    return build(cfg, NECKS)


def build_panoptic(cfg):
    return build(cfg, PANOPTIC)


def build_head(cfg):
    return build(cfg, HEADS)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    return build(cfg, DETECTORS, dict(train_cfg=train_cfg, test_cfg=test_cfg))
