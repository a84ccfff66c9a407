"""In-kernel timeline and clock of the four-wave retriever (csrc/retr_attn4.hip) from s_memtime / s_memrealtime stamps.
    make -C slotvps_amd/csrc stamp4
    SLOTVPS_LIB=slotvps_amd/libslotvps_hip_stamp4.so python tools/retr4_stamps.py [--T 5 --H 256 --W 512 --L 100]
The stamps carry an lgkmcnt(0) wait each (they drain the LDS reads in flight): read the SHARES, not the length."""
import argparse, ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd import ops, _lib
from slotvps_amd.slot_head import MaskDynamicConv

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=5)
ap.add_argument("--H", type=int, default=256)
ap.add_argument("--W", type=int, default=512)
ap.add_argument("--L", type=int, default=100)
ap.add_argument("--warm-s", type=float, default=2.0)
a = ap.parse_args()
ops.RETR_ATTN_FORM = "w4"
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = MaskDynamicConv(256).to(dev).eval()
HW = a.H * a.W
feat = torch.randn((a.T, HW, 256), device=dev).to(torch.bfloat16)
slots = torch.randn((a.T, a.L, 256), device=dev)
tabs = ops.pos_embed_sine_tables(a.H, a.W, 256, dev)
c = m._fused_consts()
with torch.no_grad():
    st = ops.retr_stats(feat, a.H, a.W, m.retr_pos_tables(tabs), c["rk"], c["rbk"], 1e-5, c["rv"], c["rbv"], 1e-5)
    fn = lambda: m.forward_fused(slots, feat, (a.H, a.W), tabs, stats=st)
    t0 = time.time()
    while time.time() - t0 < a.warm_s:                       # >= 2 s of back-to-back launches on random data
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
lib = _lib.load()
stamps = np.zeros((4, 8, 8), dtype=np.uint64)
clock = np.zeros((4096, 4), dtype=np.uint64)
lib.svps_retr4_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
rc = lib.svps_retr4_debug_read(stamps.ctypes.data_as(ctypes.c_void_p), clock.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
ok = clock[:, 2] > 0
cyc = (clock[ok, 2] - clock[ok, 0]).astype(np.float64)
rt = (clock[ok, 3] - clock[ok, 1]).astype(np.float64)          # 100 MHz ticks
ghz = cyc / rt * 0.1
print(f"workgroups {ok.sum()}: in-kernel clock median {np.median(ghz):.3f} GHz (min {ghz.min():.3f}, max {ghz.max():.3f}); "
      f"loop cycles median {np.median(cyc):.0f}, loop time median {np.median(rt) * 10:.0f} ns")
names = ["B(it) done", "logits 0-15", "logits 16-31 (+finish, conversion, DMA)", "P.f 0-8", "P.f 9-17 (+head)", "fragment prefetch"]
for w in range(4):
    print(f"--- wave {w}")
    for it in range(1, 6):
        row = stamps[w, it, :6].astype(np.int64)
        prev = stamps[w, it - 1, 5].astype(np.int64)
        if not row.all() or not prev:
            continue
        d = np.diff(np.concatenate([[prev], row]))
        print(f"it {it + 8}: " + "  ".join(f"{names[i]} +{d[i]}" for i in range(6)) + f"   = {row[5] - prev}")
    per = (stamps[w, 7, 0].astype(np.int64) - stamps[w, 1, 0].astype(np.int64)) / 6
    print("cycles per tile:", per)
