"""Per-level timing of K3 (kv_project), K4 (level_fuse) and K2 (mask_decode), warmed clocks."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd import ops, _lib, synth
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=5)
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--levels", default="32x64,64x128,128x256,256x512")
ap.add_argument("--which", default="k3,k4,k2")
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
p = synth.make_params(synth.retriever_shapes(""), 1)
P = {k: torch.from_numpy(v).to(dev) for k, v in p.items()}
wk, wv = P["to_k.weight"].to(torch.bfloat16).contiguous(), P["to_v.weight"].to(torch.bfloat16).contiguous()
wc = torch.randn((256, 384), generator=g, device=dev).mul_(0.05).to(torch.bfloat16)
bc = torch.zeros(256, device=dev)

def timeit(fn, kid, byt, name):
    t0 = time.time()
    while time.time() - t0 < 0.3:
        for _ in range(10): fn()
        torch.cuda.synchronize()
    with ops.KernelTimer() as kt:
        for _ in range(a.iters): fn()
        torch.cuda.synchronize()
        ms, n = kt.collect(kid)
    us = ms / n * 1e3
    print(f"{name}: {us:8.1f} us  {byt/us/1e3:7.1f} GB/s ({byt/us/1e3/80:.1f}% of 8 TB/s)")

for lv in a.levels.split(","):
    H, W = (int(x) for x in lv.split("x")); HW = H * W
    f = torch.randn((a.T, HW, 256), generator=g, device=dev).to(torch.bfloat16)
    tabs = ops.pos_embed_sine_tables(H, W, 256, dev)
    if "k3" in a.which:
        timeit(lambda: ops.kv_project(f, H, W, tabs, wk, P["to_k.bias"], P["norm_k.weight"], P["norm_k.bias"], 1e-5, wv,
                                      P["to_v.bias"], P["norm_v.weight"], P["norm_v.bias"], 1e-5),
               _lib.KERNEL_KV_PROJECT, a.T * HW * 1536, f"K3 {lv:>8} T={a.T}")
    if "k4" in a.which and H % 2 == 0:
        cur = torch.randn((a.T, 128, H, W), generator=g, device=dev)
        prev = torch.randn((a.T, HW // 4, 256), generator=g, device=dev).to(torch.bfloat16)
        timeit(lambda: ops.level_fuse(cur, prev, wc, bc, H, W), _lib.KERNEL_LEVEL_FUSE, a.T * HW * (512 + 512 + 128), f"K4 {lv:>8} nchw-f32 in")
        curb = torch.randn((a.T, HW, 128), generator=g, device=dev).to(torch.bfloat16)
        timeit(lambda: ops.level_fuse(curb, prev, wc, bc, H, W), _lib.KERNEL_LEVEL_FUSE, a.T * HW * (256 + 512 + 128), f"K4 {lv:>8} bf16 in    ")
    if "k2" in a.which:
        e = torch.relu(torch.randn((a.T, 100, 256), generator=g, device=dev))
        timeit(lambda: ops.mask_decode(f, e, torch.ones(256, device=dev), bc, 0.1, 0.0, want_argmax=True),
               _lib.KERNEL_MASK_DECODE, a.T * HW * (512 + 400 + 1), f"K2 {lv:>8} fp32 out + argmax")
        timeit(lambda: ops.mask_decode(f, e, torch.ones(256, device=dev), bc, 0.1, 0.0, want_argmax=True, want_logits=False),
               _lib.KERNEL_MASK_DECODE, a.T * HW * (512 + 1), f"K2 {lv:>8} argmax only      ")
