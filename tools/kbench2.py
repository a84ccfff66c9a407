"""Repeat-timing of K1 at one level to see drift / variance. python tools/kbench2.py [--reps 6]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd import ops, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--split", type=int, default=1)
ap.add_argument("--T", type=int, default=5)
ap.add_argument("--hw", type=int, default=256 * 512)
ap.add_argument("--chunks", type=int, default=0)
ap.add_argument("--copy", type=int, default=1)
ap.add_argument("--warm-ms", type=float, default=400.0)
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
ln = torch.nn.functional.layer_norm
q = ln(torch.randn((a.T, 100, 256), generator=g, device=dev), (256,)).to(torch.bfloat16)
k = ln(torch.randn((a.T, a.hw, 256), generator=g, device=dev), (256,)).to(torch.bfloat16)
v = ln(torch.randn((a.T, a.hw, 256), generator=g, device=dev), (256,)).to(torch.bfloat16)
w = torch.ones(256, device=dev); b = torch.zeros(256, device=dev)
byt = 2 * a.T * a.hw * 512
print("plan", ops.slot_attn_plan(a.T, 100, a.hw, a.chunks))
if a.copy:
    dst = torch.empty_like(k)
    for _ in range(3): dst.copy_(k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): dst.copy_(k)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"torch copy {k.numel()*2/1e6:.0f} MB: {us:.1f} us  r+w {2*k.numel()*2/us/1e3:.0f} GB/s")
t0 = time.time()
while (time.time() - t0) * 1e3 < a.warm_ms:      # clocks ramp up under sustained load: warm first
    for _ in range(20):
        ops.slot_attn(q, k, v, w, b, split_p=bool(a.split), chunks=a.chunks)
    torch.cuda.synchronize()
for rep in range(a.reps):
    with ops.KernelTimer() as kt:
        for _ in range(a.iters):
            ops.slot_attn(q, k, v, w, b, split_p=bool(a.split), chunks=a.chunks)
        torch.cuda.synchronize()
        ms, n = kt.collect(_lib.KERNEL_SLOT_ATTN)
    us = ms / n * 1e3
    print(f"rep {rep}: {us:7.1f} us  {byt/us/1e3:7.1f} GB/s ({byt/us/1e3/80:.1f}%)")
