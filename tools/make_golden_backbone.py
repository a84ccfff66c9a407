"""Golden fixture for the PyTorch backbone / neck mirrors by RUNNING the reference's own ResNet and FPN
(mmdet/models/backbones/resnet.py, mmdet/models/necks/fpn.py with mmdet/models/utils/{norm,conv_ws,conv_module}.py) in
the build container on a small input with seeded weights and BatchNorm statistics.

Stand-ins (non-arithmetic at inference): mmcv.cnn.{constant,kaiming,xavier}_init (construction-time initialisers; every
tensor is overwritten from the seed afterwards), mmcv.runner.{load_checkpoint,load_state_dict} (unused),
mmdet.core.auto_fp16 (identity), mmdet.core.utils.misc.NestedTensor (holder), mmdet.models.plugins.GeneralizedAttention and
mmdet.ops.{ContextBlock,DeformConv,ModulatedDeformConv} (not instantiated: dcn / gcb / gen_attention are None in the Slot-VPS
configs), ..registry (identity decorators). Stored: state-dict key / shape lists of ResNet-50 and FPN and their outputs."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"
R50 = dict(depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1, norm_eval=True, style="pytorch")
FPN = dict(in_channels=[256, 512, 1024, 2048], out_channels=256, num_outs=5)


def seeded_state(model, seed):
    """Deterministic weights of trained-model magnitudes: convs ~ N(0, 2 / fan_in), BN weight ~ 1, bias ~ 0.1, running
    mean ~ 0.1, running var in [0.5, 1.5]."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in model.state_dict().items():
        if not v.dtype.is_floating_point:
            out[k] = v.clone()
        elif k.endswith("running_var"):
            out[k] = 0.5 + torch.rand(v.shape, generator=g)
        elif k.endswith("running_mean"):
            out[k] = 0.1 * torch.randn(v.shape, generator=g)
        elif v.dim() == 4:
            fan_in = v.shape[1] * v.shape[2] * v.shape[3]
            out[k] = torch.randn(v.shape, generator=g) * (2.0 / fan_in) ** 0.5
        elif k.endswith("weight"):
            out[k] = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        else:
            out[k] = 0.1 * torch.randn(v.shape, generator=g)
    return out


def load_reference():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__path__ = []
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m

    class _Reg:
        def register_module(self, cls):
            return cls

    class NestedTensor:
        def __init__(self, tensors, mask):
            self.tensors, self.mask = tensors, mask

    noop = lambda *a, **k: None
    mod("mmcv"); mod("mmcv.cnn", constant_init=noop, kaiming_init=noop, xavier_init=noop)
    mod("mmcv.runner", load_checkpoint=noop, load_state_dict=noop)
    mod("mmdet"); mod("mmdet.core", auto_fp16=lambda *a, **k: (lambda f: f)); mod("mmdet.core.utils")
    mod("mmdet.core.utils.misc", NestedTensor=NestedTensor)
    mod("mmdet.models"); mod("mmdet.models.plugins", GeneralizedAttention=None)
    mod("mmdet.ops", ContextBlock=None, DeformConv=None, ModulatedDeformConv=None)
    mod("refpkg"); mod("refpkg.registry", BACKBONES=_Reg(), NECKS=_Reg())
    mod("refpkg.utils")
    norm = load("refpkg.utils.norm", "mmdet/models/utils/norm.py")
    load("refpkg.utils.conv_ws", "mmdet/models/utils/conv_ws.py")
    cm = load("refpkg.utils.conv_module", "mmdet/models/utils/conv_module.py")
    sys.modules["refpkg.utils"].__dict__.update(ConvModule=cm.ConvModule, build_conv_layer=cm.build_conv_layer,
                                                build_norm_layer=norm.build_norm_layer)
    mod("refpkg.backbones"); mod("refpkg.necks")
    resnet = load("refpkg.backbones.resnet", "mmdet/models/backbones/resnet.py")
    fpn = load("refpkg.necks.fpn", "mmdet/models/necks/fpn.py")
    return resnet.ResNet, fpn.FPN


def main():
    ResNet, FPN_ = load_reference()
    bb = ResNet(**R50)
    bb.eval()
    bb.load_state_dict(seeded_state(bb, 3))
    neck = FPN_(**FPN)
    neck.eval()
    neck.load_state_dict(seeded_state(neck, 4))
    x = torch.randn(2, 3, 64, 96, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        c = bb(x)
        p = neck(c)
    np.savez_compressed(os.path.join(GOLDEN, "backbone.npz"),
                        resnet_keys=np.array(list(bb.state_dict().keys())),
                        resnet_shapes=np.array([",".join(map(str, v.shape)) for v in bb.state_dict().values()]),
                        fpn_keys=np.array(list(neck.state_dict().keys())),
                        fpn_shapes=np.array([",".join(map(str, v.shape)) for v in neck.state_dict().values()]),
                        **{f"c{i}": o.numpy() for i, o in enumerate(c)}, **{f"p{i}": o.numpy() for i, o in enumerate(p)})
    print("resnet keys", len(bb.state_dict()), "fpn keys", len(neck.state_dict()), [tuple(o.shape) for o in p])


if __name__ == "__main__":
    main()
