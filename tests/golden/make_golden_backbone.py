"""Golden fixture for the PyTorch backbone / neck mirrors by RUNNING the reference's own ResNet and FPN
(mmdet/models/backbones/resnet.py, mmdet/models/necks/fpn.py with mmdet/models/utils/{norm,conv_ws,conv_module}.py) in
the build container on a small input with seeded weights and BatchNorm statistics.

Stand-ins (non-arithmetic at inference): mmcv.cnn.{constant,kaiming,xavier}_init (construction-time initialisers; every
tensor is overwritten from the seed afterwards), mmcv.runner.{load_checkpoint,load_state_dict} (unused),
mmdet.core.auto_fp16 (identity), mmdet.core.utils.misc.NestedTensor (holder), mmdet.models.plugins.GeneralizedAttention and
mmdet.ops.{ContextBlock,DeformConv,ModulatedDeformConv} (not instantiated: dcn / gcb / gen_attention are None in the Slot-VPS
configs), ..registry (identity decorators). Stored: state-dict key / shape lists of ResNet-50 and FPN and their outputs.

UPSNetFPN (mmdet/models/panoptic/upsnetFPN.py + mmdet/models/utils/deform_conv_with_offset.py) is run the same way with ONE
ARITHMETIC stand-in: mmdet.ops.DeformConv (a CUDA-only extension, deform_conv.py:44-45) = a module with the same `weight`
parameter whose forward is the CPU oracle's deform_conv (oracle/slotvps_oracle.py, float64). That fixture therefore pins
the STRUCTURE of the semantic tower (parameter names, layer order, GroupNorm, upsampling, prediction conv) around the
oracle's deformable convolution; the deformable convolution itself is pinned by the properties in tests/test_deform_conv.py."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLDEN = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"
R50 = dict(depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1, norm_eval=True, style="pytorch")
FPN = dict(in_channels=[256, 512, 1024, 2048], out_channels=256, num_outs=5)
UPS = dict(in_channels=32, out_channels=32, num_levels=4, num_things_classes=8, num_classes=19, ignore_label=255,
           loss_weight=0.5, return_feat_levels=4)          # the r50 config's tower at 32 channels (GroupNorm(32) still valid)


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
    sys.path.insert(0, ROOT)
    from oracle import slotvps_oracle as orc

    class OracleDeformConv(torch.nn.Module):
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                     deformable_groups=1, bias=False):
            super().__init__()
            assert not bias and groups == 1
            self.stride, self.padding, self.dilation, self.dg = stride, padding, dilation, deformable_groups
            self.weight = torch.nn.Parameter(torch.zeros(out_channels, in_channels, kernel_size, kernel_size))

        def forward(self, x, offset):
            w = self.weight.detach().double().numpy()
            out = [orc.deform_conv(x[n].detach().double().numpy(), offset[n].detach().double().numpy(), w, self.stride,
                                   self.padding, self.dilation, self.dg) for n in range(x.shape[0])]
            return torch.from_numpy(np.stack(out)).float()

    mod("mmdet.ops", ContextBlock=None, DeformConv=OracleDeformConv, ModulatedDeformConv=None)
    mod("refpkg"); mod("refpkg.registry", BACKBONES=_Reg(), NECKS=_Reg(), PANOPTIC=_Reg())
    mod("refpkg.utils")
    norm = load("refpkg.utils.norm", "mmdet/models/utils/norm.py")
    load("refpkg.utils.conv_ws", "mmdet/models/utils/conv_ws.py")
    cm = load("refpkg.utils.conv_module", "mmdet/models/utils/conv_module.py")
    dwo = load("refpkg.utils.deform_conv_with_offset", "mmdet/models/utils/deform_conv_with_offset.py")
    sys.modules["refpkg.utils"].__dict__.update(ConvModule=cm.ConvModule, build_conv_layer=cm.build_conv_layer,
                                                build_norm_layer=norm.build_norm_layer,
                                                DeformConvWithOffset=dwo.DeformConvWithOffset)
    mod("refpkg.backbones"); mod("refpkg.necks")
    resnet = load("refpkg.backbones.resnet", "mmdet/models/backbones/resnet.py")
    fpn = load("refpkg.necks.fpn", "mmdet/models/necks/fpn.py")
    mod("refpkg.panoptic")
    ups = load("refpkg.panoptic.upsnetFPN", "mmdet/models/panoptic/upsnetFPN.py")
    return resnet.ResNet, fpn.FPN, ups.UPSNetFPN


def main():
    ResNet, FPN_, UPS_ = load_reference()
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
    # semantic tower on small maps; the offset convolutions get small non-zero weights so that the sampling really deforms
    tower = UPS_(**UPS)
    tower.eval()
    st = seeded_state(tower, 6)
    for k in st:
        if "conv_offset" in k:
            st[k] = st[k] * 0.3
    tower.load_state_dict(st)
    g = torch.Generator().manual_seed(7)
    lv = [torch.randn(1, 32, 16 >> i, 24 >> i, generator=g) for i in range(4)]
    with torch.no_grad():
        up_out, score, feats = tower(lv)
    np.savez_compressed(os.path.join(GOLDEN, "semantic_tower.npz"),
                        keys=np.array(list(tower.state_dict().keys())),
                        shapes=np.array([",".join(map(str, v.shape)) for v in tower.state_dict().values()]),
                        up=up_out.numpy(), score=score.numpy(), **{f"feat{i}": f.numpy() for i, f in enumerate(feats)})
    print("tower keys", len(tower.state_dict()), tuple(up_out.shape), [tuple(f.shape) for f in feats])
    np.savez_compressed(os.path.join(GOLDEN, "backbone.npz"),
                        resnet_keys=np.array(list(bb.state_dict().keys())),
                        resnet_shapes=np.array([",".join(map(str, v.shape)) for v in bb.state_dict().values()]),
                        fpn_keys=np.array(list(neck.state_dict().keys())),
                        fpn_shapes=np.array([",".join(map(str, v.shape)) for v in neck.state_dict().values()]),
                        **{f"c{i}": o.numpy() for i, o in enumerate(c)}, **{f"p{i}": o.numpy() for i, o in enumerate(p)})
    print("resnet keys", len(bb.state_dict()), "fpn keys", len(neck.state_dict()), [tuple(o.shape) for o in p])


if __name__ == "__main__":
    main()
