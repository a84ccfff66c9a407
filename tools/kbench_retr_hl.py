"""Per-kernel timing of the reference-precision retriever (K3-HL retr_stats_hl, K1'-HL retr_attn<.., HL>) at one level size with warmed
clocks, through the library's HIP-event hooks. Timing-only ablations of K1'-HL: the separate library of `make -C slotvps_amd/csrc ablate`
(SLOTVPS_LIB=slotvps_amd/libslotvps_hip_ablate.so) with SVPS_RETR_ABLATE = 1 (DMA + barriers only), 2 (producers only), 4 (consumers only).
    python tools/kbench_retr_hl.py [--T 40] [--H 256] [--W 512] [--L 100] [--reps 3]"""
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
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--warm-ms", type=float, default=400.0)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = MaskDynamicConv(256).to(dev).eval()
m.precision = "fp16x2"
g = torch.Generator(device=dev).manual_seed(0)
HW = a.H * a.W
planes = ops.split_hl(2.0 * torch.randn((a.T, HW, 256), generator=g, device=dev))
slots = torch.randn((a.T, a.L, 256), generator=g, device=dev)
tabs = ops.pos_embed_sine_tables(a.H, a.W, 256, dev)


def once():
    with torch.no_grad():
        return m.forward_pm(slots, planes, (a.H, a.W), tabs)


t0 = time.time()
while (time.time() - t0) * 1e3 < a.warm_ms:
    for _ in range(3):
        once()
    torch.cuda.synchronize()
px = a.T * HW
for rep in range(a.reps):
    with ops.KernelTimer() as kt:
        for _ in range(a.iters):
            once()
        torch.cuda.synchronize()
        s_ms, s_n = kt.collect(_lib.KERNEL_RETR_STATS)
        a_ms, a_n = kt.collect(_lib.KERNEL_RETR_ATTN)
    su, au = s_ms / s_n * 1e3, a_ms / a_n * 1e3
    print(f"rep {rep}: retr_stats_hl {su:8.1f} us ({px * 1040 / su / 1e3:6.0f} GB/s)   retr_attn_hl {au:8.1f} us ({px * 1040 / au / 1e3:6.0f} GB/s, "
          f"{px / 32 * 4 * 101 * 32768 / au / 1e6:5.0f} TF/s executed)   abl attn={os.environ.get('SVPS_RETR_ABLATE', '0')}", flush=True)
