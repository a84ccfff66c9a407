"""Per-kernel timing of the statistics-fused retriever (K3' retr_stats, K1' retr_attn) at one level size with warmed clocks,
through the library's HIP-event hooks. Ablations: build the library with EXTRA_retr_stats=-DSVPS_STATS_ABLATE and set
SVPS_STATS_ABLATE / SVPS_RETR_ABLATE (one process per value: they are read once).
    python tools/kbench_retr.py [--T 5] [--H 256] [--W 512] [--L 100] [--reps 4]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd import ops, _lib
from slotvps_amd.slot_head import MaskDynamicConv

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=5)
ap.add_argument("--H", type=int, default=256)
ap.add_argument("--W", type=int, default=512)
ap.add_argument("--L", type=int, default=100)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--warm-ms", type=float, default=400.0)
ap.add_argument("--map-dtype", default="bf16", help="bf16 or fp16 (fp16: the kernels skip their conversion pass, as on the default path)")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = MaskDynamicConv(256).to(dev).eval()
m.precision = "bf16"          # 16-bit maps (the module default is fp16x2)
g = torch.Generator(device=dev).manual_seed(0)
HW = a.H * a.W
feat = torch.randn((a.T, HW, 256), generator=g, device=dev).to(torch.float16 if a.map_dtype == "fp16" else torch.bfloat16)
slots = torch.randn((a.T, a.L, 256), generator=g, device=dev)
tabs = ops.pos_embed_sine_tables(a.H, a.W, 256, dev)
c = m._fused_consts()


def once():
    with torch.no_grad():
        st = ops.retr_stats(feat, a.H, a.W, m.retr_pos_tables(tabs), c["rk"], c["rbk"], 1e-5, c["rv"], c["rbv"], 1e-5)
        return m.forward_fused(slots, feat, (a.H, a.W), tabs, stats=st)


t0 = time.time()
while (time.time() - t0) * 1e3 < a.warm_ms:
    for _ in range(10):
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
    print(f"rep {rep}: retr_stats {su:7.1f} us ({px * 528 / su / 1e3:6.0f} GB/s, {px * 147456 / su / 1e6:5.0f} TF/s tri)   "
          f"retr_attn {au:7.1f} us ({px * 528 / au / 1e3:6.0f} GB/s, {px * 4 * a.L * 256 / au / 1e6:5.0f} TF/s alg)   "
          f"abl stats={os.environ.get('SVPS_STATS_ABLATE', '0')} attn={os.environ.get('SVPS_RETR_ABLATE', '0')}", flush=True)
