"""VERDICT r05 item 5, the pruning question for K3-HL (csrc/retr_stats_hl.hip): which 32 x 32 blocks of the statistics factors could be
skipped at weight-pack time? For every retriever stage of the synthetic head (slotvps_amd/synth.py, the weights bench.py runs) this
builds the float64 QR factor R of the centred projection [W~ | b~] exactly as MaskDynamicConv._fused_consts does (dynamic_mask_head.py:
432-433 behind it), splits it into fp16 hi + lo, and bounds what each block can contribute to a row of R x: |block . x| <= ||block||_2 ||x_blk||.
A block may be dropped only if its bound, for ANY admissible x, stays below 2^-25 of what the row's other blocks guarantee - with nothing
known about x but a per-channel scale that means ||block||_F against the row block's norm. Runs on the CPU.
    python tools/k3hl_block_norms.py > profiles/r06/k3hl_block_norms.txt"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd import synth

cfg = synth.default_cfg() if hasattr(synth, "default_cfg") else None
shapes = synth.head_shapes(cfg) if cfg is not None else synth.head_shapes()
params = synth.make_params(shapes, 0)
names = sorted(k for k in params if k.endswith("to_k.weight") or k.endswith("to_v.weight"))
print(f"{len(names)} projection matrices (key / value of every retriever stage), synthetic weights seed 0")
worst = 1.0
for n in names:
    w = torch.from_numpy(np.asarray(params[n])).double()
    b = torch.from_numpy(np.asarray(params[n.replace("weight", "bias")])).double()
    wc = w - w.mean(dim=0, keepdim=True)
    bc = b - b.mean()
    r = torch.triu(torch.linalg.qr(torch.cat([wc, bc[:, None]], dim=1), mode="r").R[:, :256])
    hi = r.to(torch.float16).double()
    lo = (r - hi).to(torch.float16).double()
    def blocks(m):
        return m.reshape(8, 32, 8, 32).permute(0, 2, 1, 3).reshape(8, 8, -1).norm(dim=2)      # Frobenius norm of block (row block i, column block j)
    bh, bl = blocks(hi), blocks(lo)
    up = torch.triu(torch.ones(8, 8), 1).bool()
    diag = torch.eye(8).bool()
    rowblk = (bh ** 2).sum(dim=1).sqrt()                                                     # norm of a row block of R_hi
    off = (bh / rowblk[:, None])[up]
    lo_rel = (bl / rowblk[:, None])[up | diag]
    worst = min(worst, off.min().item(), lo_rel.min().item())
    print(f"{n:60s} off-diagonal R_hi blocks / their row block: min {off.min():.3f} median {off.median():.3f} max {off.max():.3f}   "
          f"R_lo blocks / row block of R_hi: min 2^{np.log2(lo_rel.min().item()):.1f} median 2^{np.log2(lo_rel.median().item()):.1f} max 2^{np.log2(lo_rel.max().item()):.1f}")
print(f"smallest relative block norm over all stages: 2^{np.log2(worst):.1f}; the skip rule of VERDICT r05 item 5 needs < 2^-25 - no block qualifies: "
      "dense random factors spread a row's norm evenly over its blocks, and R_lo is the ROUNDING of R_hi (2^-12 of it, block by block)")
