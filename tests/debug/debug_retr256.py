"""Localise errors of the fused retriever for more than 128 slots: per slot block, per frame."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from util import orc, to_bf16_t
from test_retr_fused_gpu import make_module
from slotvps_amd import ops
cuda = torch.device("cuda:0")
for (T, H, W, L, pos) in [(1, 16, 32, 256, True), (1, 9, 20, 129, False), (2, 34, 60, 200, True)]:
    m, P = make_module(cuda, 7 + L)
    rng = np.random.default_rng(L + W)
    feat = orc.round_bf16(rng.standard_normal((T, H * W, 256)).astype(np.float32))
    slots = rng.standard_normal((T, L, 256)).astype(np.float32)
    tabs = ops.pos_embed_sine_tables(H, W, 256, cuda) if pos else None
    with torch.no_grad():
        got = m.forward_fused(torch.from_numpy(slots).to(cuda), to_bf16_t(feat, cuda), (H, W), tabs).cpu().numpy()
    pm = orc.pos_embed_sine(H, W) if pos else None
    for t in range(T):
        ref = orc.retriever(slots[t], feat[t], pm, P, "", st=orc.Storage.exact(), dt=np.float64)
        err = np.abs(got[t] - ref).max(axis=1)
        print(f"T={T} {H}x{W} L={L} frame {t}: per 32-slot block max err", [f"{err[b:b + 32].max():.1e}" for b in range(0, L, 32)])
