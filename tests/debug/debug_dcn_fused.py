"""Localise errors of K7' (deformable conv without column buffer): by output channel block, pixel block, tap."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from slotvps_amd.dcn import DeformConv
cuda = torch.device("cuda:0")
for (C, O, H, W, scale) in [(64, 128, 8, 16, 0.0), (256, 256, 16, 32, 0.0), (256, 256, 16, 32, 2.5)]:
    torch.manual_seed(0)
    m = DeformConv(C, O, 3, padding=1).to(cuda)
    x = torch.randn(1, C, H, W, device=cuda)
    off = scale * torch.randn(1, 18, H, W, device=cuda)
    with torch.no_grad():
        a = m(x, off)
        m.fused = False
        b = m(x, off)
        m.fused = True
    err = (a - b).abs()[0]                      # [O, H, W]
    print(f"C={C} O={O} {H}x{W} scale={scale}: max {err.max().item():.2e}")
    print("  per out block:", [f"{err[32 * i:32 * i + 32].max().item():.1e}" for i in range(O // 32)])
    e2 = err.reshape(O, -1)
    print("  per 32-px block:", [f"{e2[:, 32 * i:32 * i + 32].max().item():.1e}" for i in range(min(8, H * W // 32))])
    # single-tap weights
    for t in range(9):
        with torch.no_grad():
            wsave = m.weight.clone()
            m.weight.zero_()
            m.weight[:, :, t // 3, t % 3] = wsave[:, :, t // 3, t % 3]
            a = m(x, off); m.fused = False; b = m(x, off); m.fused = True
            m.weight.copy_(wsave)
        print(f"  tap {t}: {(a - b).abs().max().item():.1e}", end="")
    print()
    for cc in range(C // 64):
        with torch.no_grad():
            wsave = m.weight.clone()
            m.weight.zero_()
            m.weight[:, 64 * cc:64 * cc + 64] = wsave[:, 64 * cc:64 * cc + 64]
            a = m(x, off); m.fused = False; b = m(x, off); m.fused = True
            m.weight.copy_(wsave)
        print(f"  ch-chunk {cc}: {(a - b).abs().max().item():.1e}", end="")
    print()
