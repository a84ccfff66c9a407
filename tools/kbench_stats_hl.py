"""(GPU box) Timing of the reference-precision statistics kernel (csrc/retr_stats_hl.hip) at one level size with warmed clocks.
    python tools/kbench_stats_hl.py [--T 40] [--H 256] [--W 512]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd import ops, _lib
from slotvps_amd.slot_head import MaskDynamicConv
ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=40)
ap.add_argument("--H", type=int, default=256)
ap.add_argument("--W", type=int, default=512)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = MaskDynamicConv(256).to(dev).eval()
g = torch.Generator(device=dev).manual_seed(0)
f = ops.split_hl(torch.randn((a.T, a.H * a.W, 256), generator=g, device=dev))
tabs = ops.pos_embed_sine_tables(a.H, a.W, 256, dev)
c = m._fused_consts()
tyk, txk, rbv, tiled = m.stats_hl_tables(tabs)
run = lambda: ops.retr_stats_hl(f, a.H, a.W, tyk, txk, c["rk"], c["rk_lo"], 1e-5, c["rv"], c["rv_lo"], rbv, 1e-5, tx_tiled=tiled)
t0 = time.time()
while time.time() - t0 < 0.4:
    run()
torch.cuda.synchronize()
for rep in range(3):
    with ops.KernelTimer() as kt:
        for _ in range(10):
            run()
        torch.cuda.synchronize()
        ms, n = kt.collect(_lib.KERNEL_RETR_STATS)
    us = ms / n * 1e3
    px = a.T * a.H * a.W
    print(f"rep {rep}: retr_stats_hl {us:8.1f} us  ({px * 1040 / us / 1e3:6.0f} GB/s, executed {px * 3 * 147456 / us / 1e6:5.0f} TF/s)", flush=True)
