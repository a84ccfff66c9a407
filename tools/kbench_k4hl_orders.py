"""K4-HL per level: all orders G^(m) from ONE launch (svps_level_fuse_hl_multi_fwd) against one launch per order, device time by HIP events.
    python tools/kbench_k4hl_orders.py [--T 160] [--H 1024] [--W 2048]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd import synth
from slotvps_amd.clip import build_r50_head

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=160)
ap.add_argument("--H", type=int, default=1024)
ap.add_argument("--W", type=int, default=2048)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0")
head = build_r50_head().to(dev).eval()
head.set_mode("fp16x2")
sizes = synth.level_sizes(a.H, a.W)
g = torch.Generator(device=dev).manual_seed(0)
feats = [torch.randn((a.T, 128, h, w), generator=g, device=dev) for (h, w) in sizes]


def run(one):
    head.fuse_orders_in_one_launch = one
    times = []
    with torch.no_grad():
        prev = None
        for i, (h, w) in enumerate(sizes):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            prev = head.fuse_level(feats[i], prev, (h, w), last=i == 3)
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
    return times


for _ in range(3):
    run(True); run(False)
for rep in range(a.reps):
    for one in (True, False):
        t = run(one)
        print(f"rep {rep} one_launch={one}: " + "  ".join(f"level {i} ({h}x{w}) {x:7.3f} ms" for i, ((h, w), x) in enumerate(zip(sizes, t))) + f"   total {sum(t):7.3f} ms", flush=True)
