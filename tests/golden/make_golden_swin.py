"""Golden fixture for the Swin backbone by RUNNING the reference's module
(mmdet/models/backbones/swin_transformer.py) in the build container on a small configuration with seeded weights.

Stand-ins (non-arithmetic): timm.models.layers.{DropPath (identity: eval mode), to_2tuple, trunc_normal_ =
torch.nn.init.trunc_normal_}, mmcv.runner.load_checkpoint (unused), ..registry.BACKBONES (identity decorator).
Stored: the state-dict key / shape list of the Swin-L configuration (names only) and, for a small configuration
(embed 32, depths 2-2-2-2, window 7, 3x70x91 input: padding, shifted windows and odd merges all occur), the four
output maps. Tests rebuild the weights from the same seed."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLDEN = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"
SMALL = dict(embed_dim=32, depths=[2, 2, 2, 2], num_heads=[2, 4, 8, 16], window_size=7, mlp_ratio=4., qkv_bias=True,
             qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.5, ape=False, patch_norm=True,
             out_indices=(0, 1, 2, 3), use_checkpoint=False)
LARGE = dict(SMALL, embed_dim=192, depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48])


def seeded_state(model, seed):
    g = torch.Generator().manual_seed(seed)
    sd = model.state_dict()
    out = {}
    for k, v in sd.items():
        if v.dtype.is_floating_point:
            scale = 0.5 if "relative_position_bias_table" in k else (1.0 if k.endswith("norm.weight") or ".norm" in k and k.endswith("weight") else 0.08)
            t = torch.randn(v.shape, generator=g) * scale
            if k.endswith("weight") and v.dim() == 1:
                t = 1.0 + 0.2 * torch.randn(v.shape, generator=g)      # LayerNorm weights around 1
            out[k] = t
        else:
            out[k] = v.clone()
    return out


def load_reference():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__path__ = []
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Reg:
        def register_module(self, cls):
            return cls

    mod("timm"); mod("timm.models")
    mod("timm.models.layers", DropPath=lambda p=0.: torch.nn.Identity(), to_2tuple=lambda x: x if isinstance(x, tuple) else (x, x),
        trunc_normal_=torch.nn.init.trunc_normal_)
    mod("mmcv"); mod("mmcv.runner", load_checkpoint=None)
    mod("refpkg"); mod("refpkg.registry", BACKBONES=_Reg()); mod("refpkg.backbones")
    spec = importlib.util.spec_from_file_location("refpkg.backbones.swin_transformer",
                                                  os.path.join(REF, "mmdet/models/backbones/swin_transformer.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    ref = load_reference()
    big = ref.SwinTransformer(**LARGE)
    keys = [(k, tuple(v.shape)) for k, v in big.state_dict().items()]
    small = ref.SwinTransformer(**SMALL)
    small.eval()            # the reference's train() override returns None
    small.load_state_dict(seeded_state(small, 7))
    x = torch.randn(2, 3, 70, 91, generator=torch.Generator().manual_seed(8))
    with torch.no_grad():
        outs = small(x)
    np.savez_compressed(os.path.join(GOLDEN, "swin.npz"), large_keys=np.array([k for k, _ in keys]),
                        large_shapes=np.array([",".join(map(str, s)) for _, s in keys]),
                        **{f"out{i}": o.numpy() for i, o in enumerate(outs)})
    print("keys", len(keys), "outs", [tuple(o.shape) for o in outs])


if __name__ == "__main__":
    main()
