"""Kernel-level micro-benchmark of the HIP path (device time from HIP events on the launch stream).
Usage: python tools/kbench.py [--T 5] [--L 100] [--iters 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd import ops, _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, default=5)
    ap.add_argument("--L", type=int, default=100)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--levels", type=str, default="32x64,64x128,128x256,256x512")
    ap.add_argument("--chunks", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    w = torch.ones(256, device=dev)
    b = torch.zeros(256, device=dev)
    ln = torch.nn.functional.layer_norm
    for lv in a.levels.split(","):
        H, W = (int(x) for x in lv.split("x"))
        HW = H * W
        q = ln(torch.randn((a.T, a.L, 256), generator=g, device=dev), (256,)).to(torch.bfloat16)
        k = ln(torch.randn((a.T, HW, 256), generator=g, device=dev), (256,)).to(torch.bfloat16)
        v = ln(torch.randn((a.T, HW, 256), generator=g, device=dev), (256,)).to(torch.bfloat16)
        e = torch.relu(torch.randn((a.T, a.L, 256), generator=g, device=dev))
        plan = ops.slot_attn_plan(a.T, a.L, HW, a.chunks)
        for split in (True, False):
            for _ in range(3):
                ops.slot_attn(q, k, v, w, b, split_p=split, chunks=a.chunks)
            torch.cuda.synchronize()
            with ops.KernelTimer() as kt:
                for _ in range(a.iters):
                    ops.slot_attn(q, k, v, w, b, split_p=split, chunks=a.chunks)
                torch.cuda.synchronize()
                ms, n = kt.collect(_lib.KERNEL_SLOT_ATTN)
                ms2, n2 = kt.collect(_lib.KERNEL_SLOT_ATTN_FINISH)
            byt = 2 * a.T * HW * 256 * 2 + 2 * a.T * a.L * 256 * 2
            us = ms / n * 1e3
            print(f"K1 {lv:>8} T={a.T} L={a.L} split={int(split)} plan={plan}: {us:8.1f} us  "
                  f"{byt / us / 1e3:7.1f} GB/s ({byt / us / 1e3 / 8000 * 100:4.1f}% of 8 TB/s)  finish {ms2 / n2 * 1e3:6.1f} us")
        for _ in range(3):
            ops.mask_decode(k, e, w, b, 0.1, 0.0)
        torch.cuda.synchronize()
        with ops.KernelTimer() as kt:
            for _ in range(a.iters):
                ops.mask_decode(k, e, w, b, 0.1, 0.0)
            torch.cuda.synchronize()
            ms, n = kt.collect(_lib.KERNEL_MASK_DECODE)
        byt = a.T * HW * 512 + a.T * a.L * HW * 4
        us = ms / n * 1e3
        print(f"K2 {lv:>8} T={a.T} L={a.L} fp32-out: {us:8.1f} us  {byt / us / 1e3:7.1f} GB/s ({byt / us / 1e3 / 8000 * 100:4.1f}%)")


if __name__ == "__main__":
    main()
