"""Per-kernel timing of the fused retriever's precision form (K3t statistics with fp16 hi + lo factors, K1' with P * rstd_v as fp16
hi + lo) against the default form, at one level size.   python tools/kbench_tight.py [--T 40 --H 256 --W 512 --L 100]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd import ops, _lib
from slotvps_amd.slot_head import MaskDynamicConv

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=40)
ap.add_argument("--H", type=int, default=256)
ap.add_argument("--W", type=int, default=512)
ap.add_argument("--L", type=int, default=100)
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = MaskDynamicConv(256).to(dev).eval()
g = torch.Generator(device=dev).manual_seed(0)
feat = torch.randn((a.T, a.H * a.W, 256), generator=g, device=dev).to(torch.bfloat16)
slots = torch.randn((a.T, a.L, 256), generator=g, device=dev)
tabs = ops.pos_embed_sine_tables(a.H, a.W, 256, dev)
for tight in (False, True):
    m.tight_stats = tight
    with torch.no_grad():
        t0 = time.time()
        while time.time() - t0 < 0.5:
            m.forward_fused(slots, feat, (a.H, a.W), tabs)
            torch.cuda.synchronize()
        with ops.KernelTimer() as kt:
            for _ in range(a.iters):
                m.forward_fused(slots, feat, (a.H, a.W), tabs)
            torch.cuda.synchronize()
            s_ms, s_n = kt.collect(_lib.KERNEL_RETR_STATS)
            a_ms, a_n = kt.collect(_lib.KERNEL_RETR_ATTN)
    print(f"{'precision form' if tight else 'default form  '}: statistics {s_ms / s_n * 1e3:8.1f} us, retriever {a_ms / a_n * 1e3:8.1f} us  (T={a.T} {a.H}x{a.W} L={a.L})")
