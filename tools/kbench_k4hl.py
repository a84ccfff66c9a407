"""(GPU box) Timing of the reference-precision level fusion (csrc/level_fuse_hl.hip) at one level size: with / without taps, with / without
the fp32 copy, and the coarse product on K8.    python tools/kbench_k4hl.py [--T 40] [--H 256] [--W 512]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd import ops, _lib
ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=40)
ap.add_argument("--H", type=int, default=256)
ap.add_argument("--W", type=int, default=512)
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
cur = torch.randn((a.T, 128, a.H, a.W), generator=g, device=dev)
prev = torch.randn((a.T, (a.H // 2) * (a.W // 2), 256), generator=g, device=dev)
wts = ops.level_fuse_hl_weights(torch.randn((256, 384), generator=g, device=dev) * 0.05)
bc = torch.randn(256, generator=g, device=dev)
px = a.T * a.H * a.W


def timed(fn, name, n=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"{name:46s} {us:9.1f} us   {px / us / 1e3:7.2f} Gpx/s", flush=True)


t0 = time.time()
while time.time() - t0 < 0.3:
    ops.level_fuse_hl(cur, prev, wts, bc, a.H, a.W)
torch.cuda.synchronize()
timed(lambda: ops.level_fuse_hl(cur, prev, wts, bc, a.H, a.W), "taps (K8 coarse product + kernel)")
timed(lambda: ops.level_fuse_hl(cur, prev, wts, bc, a.H, a.W, want_f32=True), "taps + fp32 copy")
timed(lambda: ops.level_fuse_hl(cur, None, wts, bc, a.H, a.W), "no taps (level-0 form at this size)")
timed(lambda: ops.slot_gemm(prev.view(-1, 256), wts["wa_pack"]), "coarse product alone (K8 fp16 split)")
