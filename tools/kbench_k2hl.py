"""(GPU box) Timing of the reference-precision mask decode (K2-HL, csrc/mask_decode.hip) at one level size.
    python tools/kbench_k2hl.py [--T 40] [--H 256] [--W 512] [--L 100]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd import ops, _lib
ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=40)
ap.add_argument("--H", type=int, default=256)
ap.add_argument("--W", type=int, default=512)
ap.add_argument("--L", type=int, default=100)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
HW = a.H * a.W
planes = ops.split_hl(2.0 * torch.randn((a.T, HW, 256), generator=g, device=dev))
e = torch.randn((a.T, a.L, 256), generator=g, device=dev)
sc = torch.rand(256, generator=g, device=dev) + 0.5
sh = torch.randn(256, generator=g, device=dev) * 0.1
fn = lambda: ops.mask_decode_hl(planes, e, sc, sh, 0.1, 0.0, want_argmax=True)
t0 = time.time()
while time.time() - t0 < 0.4:
    fn()
    torch.cuda.synchronize()
px = a.T * HW
for rep in range(a.reps):
    with ops.KernelTimer() as kt:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        ms, n = kt.collect(_lib.KERNEL_MASK_DECODE)
    us = ms / n * 1e3
    print(f"rep {rep}: mask_decode_hl {us:8.1f} us  {px * (1024 + 4 * a.L + 1) / us / 1e3:6.0f} GB/s algorithmic", flush=True)
