"""In-kernel timeline of K1'-HL32 (retr_attn_hl32.hip) from s_memtime stamps: producer 0 and consumer 0 of one workgroup, iterations 8 .. 15.
    make -C slotvps_amd/csrc stamp [R32_EXTRA=-DSVPS_RETR_HL32_SPLIT=4]
    SLOTVPS_LIB=slotvps_amd/libslotvps_hip_stamp.so python tools/retr32_stamps.py [--T 40 --H 256 --W 512 --L 100]"""
import argparse, ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd import ops, _lib
from slotvps_amd.slot_head import MaskDynamicConv

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=40)
ap.add_argument("--H", type=int, default=256)
ap.add_argument("--W", type=int, default=512)
ap.add_argument("--L", type=int, default=100)
ap.add_argument("--warm-s", type=float, default=1.0)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = MaskDynamicConv(256).to(dev).eval()
m.precision = "fp16x2"
HW = a.H * a.W
planes = ops.split_hl(2.0 * torch.randn((a.T, HW, 256), device=dev))
slots = torch.randn((a.T, a.L, 256), device=dev)
tabs = ops.pos_embed_sine_tables(a.H, a.W, 256, dev)
with torch.no_grad():
    t0 = time.time()
    while time.time() - t0 < a.warm_s:
        for _ in range(5):
            m.forward_pm(slots, planes, (a.H, a.W), tabs)
        torch.cuda.synchronize()
lib = _lib.load()
st = np.zeros((2, 8, 8), dtype=np.uint64)
lib.svps_retr32_debug_read.argtypes = [ctypes.c_void_p]
assert lib.svps_retr32_debug_read(st.ctypes.data_as(ctypes.c_void_p)) == 0
st = st.astype(np.int64)
names = [["top (B1 passed)", "chain end", "head end (stats written)", "B1 passed", "finish end (P stored, flag set)"],
         ["top (B1 passed)", "rest of the tile before", "flag seen, P fragments, DMA issued", "first steps end", "batch landed"]]
for role in (0, 1):
    print("--- producer 0" if role == 0 else "--- consumer 0")
    for it in range(1, 7):
        row = st[role, it, :5]
        if not row.all():
            continue
        prev = st[role, it - 1, 4]
        d = np.diff(np.concatenate([[prev], row]))
        print(f"it {it + 8}: " + "  ".join(f"{names[role][k]} +{d[k]}" for k in range(5)))
print("cycles per tile (producer 0, s_memtime ticks of 100 MHz x ... see retr_stamps.py): ", (st[0, 7, 0] - st[0, 1, 0]) / 6)
print("producer top - consumer top per iteration:", [int(st[0, i, 0] - st[1, i, 0]) for i in range(1, 7)])
