"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE's own modules.

Runs only in the build container (needs /root/reference; the GPU box never sees it). It loads
    mmdet/models/detectors/dynamic_mask_head.py   (MultiScaleDynamicMaskHead, MaskRCNNHead,
                                                   MaskDynamicConv, TemporalSlotsHead, SlotsDynamicConv)
    mmdet/models/detectors/position_encoding.py   (PositionEmbeddingSine)
by file path under a synthetic package whose only members are three non-arithmetic stand-ins for
imports the container cannot satisfy:
    ..registry.HEADS         -> identity register_module decorator
    ..utils.ConvModule       -> nn.Conv2d(i, o, 1, bias=True) held as `.conv` (what ConvModule builds for
                                activation=None, no norm: conv_module.py:95-97,135)
    timm.models.layers.DropPath -> never constructed (drop_path = 0)
    mmdet.core.utils.misc    -> a two-field NestedTensor holder (the real file imports torchvision)
Nothing from the reference is written into the repo: fixtures hold seeds, small inputs and the
reference's OUTPUTS only. Weights and inputs are regenerated from tests/synth.py by the tests.

generate_final_outputs (vps_temporal_slots.py:144-160) cannot be imported (its module needs mmcv);
its fixture is produced by executing the torch ops of those six lines here (eval BatchNorm2d,
F.normalize, einsum, BatchNorm2d(1) with slots as batch).

Usage: python tests/golden/make_golden.py [--ref /root/reference]
"""
import argparse
import importlib.util
import os
import sys
import types

import numpy as np
import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_reference(ref):
    det = os.path.join(ref, "mmdet", "models", "detectors")

    def pkg(name):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
        return m

    pkg("refpkg")
    pkg("refpkg.models")
    pkg("refpkg.models.detectors")
    reg = types.ModuleType("refpkg.models.registry")

    class _Reg:
        def register_module(self, cls):
            return cls
    reg.HEADS = _Reg()
    sys.modules["refpkg.models.registry"] = reg
    utils = types.ModuleType("refpkg.models.utils")

    class ConvModule(nn.Module):
        def __init__(self, i, o, k, padding=0, activation=None):
            super().__init__()
            assert k == 1 and padding == 0 and activation is None
            self.conv = nn.Conv2d(i, o, 1, bias=True)

        def forward(self, x):
            return self.conv(x)
    utils.ConvModule = ConvModule
    sys.modules["refpkg.models.utils"] = utils
    timm = types.ModuleType("timm")
    tm = types.ModuleType("timm.models")
    tl = types.ModuleType("timm.models.layers")
    tl.DropPath = None
    sys.modules.update({"timm": timm, "timm.models": tm, "timm.models.layers": tl})
    misc = types.ModuleType("mmdet.core.utils.misc")

    class NestedTensor:
        def __init__(self, tensors, mask):
            self.tensors, self.mask = tensors, mask
    misc.NestedTensor = NestedTensor
    misc.trunc_normal_ = None
    for n in ("mmdet", "mmdet.core", "mmdet.core.utils"):
        if n not in sys.modules:
            pkg(n)
    sys.modules["mmdet.core.utils.misc"] = misc

    def load(name, fn):
        spec = importlib.util.spec_from_file_location(f"refpkg.models.detectors.{name}", os.path.join(det, fn))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[spec.name] = mod
        spec.loader.exec_module(mod)
        return mod
    return load("dynamic_mask_head", "dynamic_mask_head.py"), load("position_encoding", "position_encoding.py"), NestedTensor


def load_state(module, params, prefix=""):
    sd = module.state_dict()
    new = {}
    for k in sd:
        new[k] = torch.from_numpy(params[prefix + k]).reshape(sd[k].shape)
    module.load_state_dict(new, strict=True)


def t2n(x):
    return x.detach().cpu().numpy().astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    a = ap.parse_args()
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    dmh, pe, NestedTensor = load_reference(a.ref)
    os.makedirs(GOLDEN, exist_ok=True)
    D = 256

    # ---- a7: sine position embedding ---------------------------------------------------------------
    pos_mod = pe.PositionEmbeddingSine(128, normalize=True)
    out = {}
    for (H, W) in [(2, 4), (16, 32), (33, 65), (34, 60)]:
        x = torch.zeros(1, 128, H, W)
        p = pos_mod(NestedTensor(x, torch.zeros(1, H, W, dtype=torch.bool)))       # [1, 256, H, W]
        out[f"pos_{H}x{W}"] = t2n(p[0].permute(1, 2, 0).reshape(H * W, D))          # pixel-major view
    np.savez_compressed(os.path.join(GOLDEN, "pos_embed_sine.npz"), **out)

    def pos_nchw(H, W):
        return pos_mod(NestedTensor(torch.zeros(1, 128, H, W), torch.zeros(1, H, W, dtype=torch.bool)))

    # ---- a1: MaskDynamicConv (slot <-> pixel retriever) ---------------------------------------------
    out = {}
    for tag, (L, H, W, seed) in {"L100_16x32": (100, 16, 32, 101), "L37_9x13": (37, 9, 13, 102),
                                 "L200_6x10": (200, 6, 10, 103)}.items():
        params = synth.make_params(synth.retriever_shapes(""), seed)
        m = dmh.MaskDynamicConv(dh_dim=D, softmax_dim="slots").eval()
        load_state(m, params)
        rng = np.random.default_rng(seed + 1000)
        slots = rng.standard_normal((1, L, D)).astype(np.float32)
        feat = synth.smooth_features(rng, D, H, W)[None]
        y = m(torch.from_numpy(slots), torch.from_numpy(feat), pos_nchw(H, W))
        out[f"{tag}_out"] = t2n(y[0])
        out[f"{tag}_meta"] = np.array([L, H, W, seed], dtype=np.int64)
    np.savez_compressed(os.path.join(GOLDEN, "retriever.npz"), **out)

    # ---- a2: SlotsDynamicConv + TemporalSlotsHead ------------------------------------------------------
    out = {}
    for tag, (N, ff, act, seed) in {"N200_relu": (200, 1024, "relu", 201), "N150_gelu": (150, 1024, "gelu", 202)}.items():
        params = synth.make_params(synth.temporal_shapes("", ff), seed)
        m = dmh.TemporalSlotsHead(d_model=D, dim_feedforward=ff, dropout=0.0, activation=act).eval()
        load_state(m, params)
        rng = np.random.default_rng(seed + 1000)
        S = rng.standard_normal((N, D)).astype(np.float32)
        ts = torch.from_numpy(S)
        out[f"{tag}_out"] = t2n(m(features=ts, mask_query=ts, pos=None, query_pos=None))
        out[f"{tag}_inner"] = t2n(m.inst_interact(ts[None], ts[None], None)[0])
        out[f"{tag}_meta"] = np.array([N, ff, seed], dtype=np.int64)
    np.savez_compressed(os.path.join(GOLDEN, "temporal_head.npz"), **out)

    # ---- a3-a6: the whole multi-scale head on a small clip -------------------------------------------------
    cfg = synth.R50_HEAD_CFG
    head = dmh.MultiScaleDynamicMaskHead(
        dh_dim=D, num_classes=cfg["num_classes"], dim_feedforward=cfg["dim_feedforward"], nhead=cfg["nhead"],
        dropout=0.0, activation=cfg["activation"], dh_num_heads=7, per_dh_num_heads=list(cfg["per_dh_num_heads"]),
        feat_num_levels=4, merge_operation="concat", trans_in_dim=cfg["trans_in_dim"], return_intermediate=True,
        use_focal=True, prior_prob=0.01, num_cls=cfg["num_cls"], num_reg=cfg["num_reg"], drop_path=0.,
        temporal_query_attention_config=dict(d_model=D, dim_feedforward=cfg["temporal_dim_feedforward"], dropout=0.0,
                                             activation=cfg["temporal_activation"], softmax_dim="slots", drop_path=0.),
        apply_temporal_query_atten_stages=list(cfg["apply_temporal_query_atten_stages"])).eval()
    shapes = synth.head_shapes(cfg)
    assert set(shapes) == set(head.state_dict()), sorted(set(shapes) ^ set(head.state_dict()))[:10]
    for k, v in head.state_dict().items():
        assert tuple(v.shape) == tuple(shapes[k]), (k, v.shape, shapes[k])
    out = {}
    for tag, (T, H, W, L, seed) in {"T2_64x128": (2, 64, 128, 100, 301), "T3_64x64": (3, 64, 64, 37, 302)}.items():
        params = synth.make_params(shapes, seed)
        load_state(head, params)
        feats = synth.make_clip_features(seed + 1, T, H, W)
        slots = synth.make_slots(seed + 2, L)
        sizes = synth.level_sizes(H, W)
        pos = [pos_nchw(h, w) for (h, w) in sizes]
        features = [[torch.from_numpy(f[None]) for f in feats[t]] for t in range(T)]
        init = [torch.from_numpy(slots.copy()) for _ in range(T)]
        logits, embeds, fused = head(features=features, init_masks=init, pad_mask=None,
                                     pos=[pos for _ in range(T)], query_pos=None)
        for t in range(T):
            out[f"{tag}_logits_{t}"] = t2n(logits[t][:, 0])          # [7, L, 20]
            out[f"{tag}_embeds_{t}"] = t2n(embeds[t][:, 0])          # [7, L, 256]
            out[f"{tag}_fused3_{t}"] = t2n(fused[t][3][0].permute(1, 2, 0).reshape(-1, D))   # finest level, pixel-major
            out[f"{tag}_fused0_{t}"] = t2n(fused[t][0][0].permute(1, 2, 0).reshape(-1, D))
        out[f"{tag}_meta"] = np.array([T, H, W, L, seed], dtype=np.int64)

        # ---- a8: generate_final_outputs lines 144-160 on the head's own outputs (frame T-1) -----------
        rng = np.random.default_rng(seed + 3)
        feat_bn = nn.BatchNorm2d(D).eval()
        fg_bn = nn.BatchNorm2d(1).eval()
        feat_bn.weight.data = torch.from_numpy(rng.uniform(0.5, 1.5, D).astype(np.float32))
        feat_bn.bias.data = torch.from_numpy((0.1 * rng.standard_normal(D)).astype(np.float32))
        feat_bn.running_mean.data = torch.from_numpy((0.2 * rng.standard_normal(D)).astype(np.float32))
        feat_bn.running_var.data = torch.from_numpy(rng.uniform(0.5, 2.0, D).astype(np.float32))
        fg_bn.weight.data.fill_(0.1)
        fg_bn.bias.data.fill_(0.03)
        fg_bn.running_mean.data.fill_(0.2)
        fg_bn.running_var.data.fill_(1.7)
        f = fused[T - 1][3]                                           # [1, 256, h, w]
        g = torch.nn.functional.normalize(feat_bn(f), p=2, dim=1)     # :146-147
        m = torch.einsum("nchw,nlc->nlhw", g, embeds[T - 1][-1])      # :149
        m = fg_bn(m.permute(1, 0, 2, 3)).permute(1, 0, 2, 3)          # :153-154
        out[f"{tag}_mask"] = t2n(m[0].reshape(m.shape[1], -1))        # [L, HW]
        out[f"{tag}_bn"] = np.stack([t2n(feat_bn.weight), t2n(feat_bn.bias), t2n(feat_bn.running_mean),
                                     t2n(feat_bn.running_var)])
        out[f"{tag}_fg"] = np.array([0.1, 0.03, 0.2, 1.7], dtype=np.float32)
    np.savez_compressed(os.path.join(GOLDEN, "head_small.npz"), **out)
    for fn in sorted(os.listdir(GOLDEN)):
        print(fn, os.path.getsize(os.path.join(GOLDEN, fn)) // 1024, "KiB")


if __name__ == "__main__":
    main()
