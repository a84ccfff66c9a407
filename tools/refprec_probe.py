"""(GPU box) Where does a precision mode's distance from the REFERENCE's fp32 outputs come from? Every stage of the head teacher-forced on
the reference's own incoming slots (tests/golden/head_small.npz), per mode / variant, next to the free-running distances.
    python tools/refprec_probe.py [--modes fp32 fp16x2 fp16x2:libgemm ...]
Variants of fp16x2 (switches on the module tree, for elimination): libgemm = dense layers in library fp32; bf16k9 = K9 with the bf16 split."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402
from oracle import slotvps_oracle as orc  # noqa: E402
from slotvps_amd import ops  # noqa: E402
from test_head_gpu import build_head  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--modes", nargs="+", default=["fp32", "fp16x2", "fp16x2:libgemm"])
a = ap.parse_args()
dev = torch.device("cuda:0")
z = np.load(os.path.join(ROOT, "tests", "golden", "head_small.npz"))
cfg = dict(orc.DEFAULT_CFG)
for tag in ("T2_64x128", "T3_64x64"):
    T, H, W, L, seed = (int(x) for x in z[f"{tag}_meta"])
    params = synth.make_params(synth.head_shapes(), seed)
    feats = synth.make_clip_features(seed + 1, T, H, W)
    slots = synth.make_slots(seed + 2, L)
    sizes = synth.level_sizes(H, W)
    for mode in a.modes:
        prec, _, var = mode.partition(":")
        head = build_head(dev, params).set_mode(prec)
        if var == "libgemm":
            head.set_slot_gemm(False)
        saved_bgemm = ops.bgemm
        if var == "bf16k9":
            ops.bgemm = lambda *x, **k: saved_bgemm(*x, **dict(k, split="bf16"))
        if var == "f64k9":
            def _bg(a_, b_, bias=None, alpha=1.0, out=None, split=None):
                a3 = a_ if a_.dim() == 3 else a_.unsqueeze(0)
                b3 = b_ if b_.dim() == 3 else b_.unsqueeze(0)
                r = alpha * torch.matmul(a3.double(), b3.double().transpose(1, 2))
                if bias is not None:
                    r = r + (bias if bias.dim() == 2 else bias.unsqueeze(0)).double()[:, None, :]
                r = r.float()
                if out is not None:
                    out.copy_(r)
                    return out
                return r
            ops.bgemm = _bg
        try:
            with torch.no_grad():
                tf = [torch.from_numpy(np.stack([feats[t][i] for t in range(T)])).to(dev) for i in range(4)]
                pos_tabs = [ops.pos_embed_sine_tables(h, w, 256, dev) for (h, w) in sizes]
                logits, embeds, fused = head.forward_clip(tf, torch.from_numpy(slots).to(dev), pos_tabs)
                tf_e, tf_l, sidx = [], [], 0
                for lvl, n in enumerate(cfg["per_level_stages"]):
                    h, w = sizes[lvl]
                    for j in range(n):
                        s_in = np.stack([slots.astype(np.float32) if sidx == 0 else z[f"{tag}_embeds_{t}"][sidx - 1] for t in range(T)])
                        stage = getattr(head, f"head_series_{lvl}")[j]
                        lg, em = stage.forward_pm(torch.from_numpy(s_in).to(dev), fused[lvl], (h, w), pos_tabs[lvl], sidx in cfg["temporal_stages"], 1)
                        tf_e.append(max(np.abs(em[t].cpu().numpy() - z[f"{tag}_embeds_{t}"][sidx]).max() for t in range(T)))
                        tf_l.append(max(np.abs(lg[t].cpu().numpy() - z[f"{tag}_logits_{t}"][sidx]).max() for t in range(T)))
                        sidx += 1
            e = embeds.cpu().numpy()
            free = [max(np.abs(e[s, t] - z[f"{tag}_embeds_{t}"][s]).max() for t in range(T)) for s in range(7)]
            print(f"[{tag}] {mode:16s} teacher-forced embeds " + " ".join(f"{x:.1e}" for x in tf_e) + " | logits " + " ".join(f"{x:.1e}" for x in tf_l)
                  + " | free-running embeds " + " ".join(f"{x:.1e}" for x in free), flush=True)
        finally:
            ops.bgemm = saved_bgemm
