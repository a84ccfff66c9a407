"""In-kernel timeline and clock of K1' (retr_attn) from s_memtime / s_memrealtime stamps. Needs the diagnostic library:
    make -C slotvps_amd/csrc stamp
    SLOTVPS_LIB=slotvps_amd/libslotvps_hip_stamp.so python tools/retr_stamps.py [--T 5 --H 256 --W 512 --L 100]"""
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
ap.add_argument("--map-dtype", default="fp16", help="fp16 (as on the default path: no conversion pass) or bf16")
ap.add_argument("--hl", type=int, default=0, help="1: the fp16x2 form (hi / lo planes, K3-HL + K1'-HL)")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = MaskDynamicConv(256).to(dev).eval()
m.precision = "bf16"          # 16-bit maps (the module default is fp16x2)
HW = a.H * a.W
feat = torch.randn((a.T, HW, 256), device=dev).to(torch.float16 if a.map_dtype == "fp16" else torch.bfloat16)
slots = torch.randn((a.T, a.L, 256), device=dev)
tabs = ops.pos_embed_sine_tables(a.H, a.W, 256, dev)
c = m._fused_consts()
if a.hl:
    m.precision = "fp16x2"
    planes = ops.split_hl(2.0 * torch.randn((a.T, HW, 256), device=dev))
with torch.no_grad():
    st = None if a.hl else ops.retr_stats(feat, a.H, a.W, m.retr_pos_tables(tabs), c["rk"], c["rbk"], 1e-5, c["rv"], c["rbv"], 1e-5)
    fn = (lambda: m.forward_pm(slots, planes, (a.H, a.W), tabs)) if a.hl else (lambda: m.forward_fused(slots, feat, (a.H, a.W), tabs, stats=st))
    t0 = time.time()
    while time.time() - t0 < a.warm_s:                       # >= 2 s of back-to-back launches on random data
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
lib = _lib.load()
stamps = np.zeros((2, 8, 8), dtype=np.uint64)
clock = np.zeros((4096, 4), dtype=np.uint64)
lib.svps_retr_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
rc = lib.svps_retr_debug_read(stamps.ctypes.data_as(ctypes.c_void_p), clock.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
ok = clock[:, 2] > 0
cyc = (clock[ok, 2] - clock[ok, 0]).astype(np.float64)
rt = (clock[ok, 3] - clock[ok, 1]).astype(np.float64)          # 100 MHz ticks
ghz = cyc / rt * 0.1
print(f"workgroups {ok.sum()}: in-kernel clock median {np.median(ghz):.3f} GHz (min {ghz.min():.3f}, max {ghz.max():.3f}); "
      f"loop cycles median {np.median(cyc):.0f}, loop time median {np.median(rt) * 10:.0f} ns")
names = [{0: "pre B", 1: "B done", 2: "MFMA chain + finish of tile it-1", 6: "logits + max", 3: "exp + sum + stats write",
          5: "next tile's position terms (end)"},
         {0: "pre wait_vm", 1: "DMA landed", 2: "B done", 3: "DMA issued", 5: "36 MFMA (end)"}]
order = [[0, 1, 2, 6, 3, 5], [0, 1, 2, 3, 5]]
for role in (0, 1):
    print("--- producer 0" if role == 0 else "--- consumer 0")
    for it in range(1, 5):
        row = stamps[role, it, order[role]].astype(np.int64)
        if not row.all():
            continue
        d = np.diff(np.concatenate([[stamps[role, it - 1, 5].astype(np.int64)], row]))
        print(f"it {it + 8}: " + "  ".join(f"{names[role][k]} +{d[i]}" for i, k in enumerate(order[role])))
per = (stamps[0, 7, 0].astype(np.int64) - stamps[0, 1, 0].astype(np.int64)) / 6
print("cycles per tile (producer 0):", per)

# ---- K3' (retr_stats): key wave 0 and value wave 0 of one workgroup
sst = np.zeros((2, 8, 8), dtype=np.uint64)
lib.svps_stats_debug_read.argtypes = [ctypes.c_void_p]
if lib.svps_stats_debug_read(sst.ctypes.data_as(ctypes.c_void_p)) == 0:
    snames = {0: "loop top", 1: "barrier A done", 2: "first phase (key: heavy, value: light) end", 3: "barrier B done",
              7: "second phase (key: light, value: heavy) end", 6: "MFMAs done", 4: "own pieces landed", 5: "converted"}
    for role in (0, 1):
        print("--- K3' key wave 0" if role == 0 else "--- K3' value wave 0")
        for it in range(1, 5):
            pts = [k for k in range(8) if sst[role, it, k]]
            pts.sort(key=lambda k: int(sst[role, it, k]))                       # in time order (the schedule decides it)
            row = sst[role, it, pts].astype(np.int64)
            prev = max(int(x) for x in sst[role, it - 1] if x)
            d = np.diff(np.concatenate([[prev], row]))
            print(f"it {it + 8}: " + "  ".join(f"{snames[k]} +{d[i]}" for i, k in enumerate(pts)))
    print("cycles per tile (K3' key wave 0):", (sst[0, 7, 0].astype(np.int64) - sst[0, 1, 0].astype(np.int64)) / 6)
