"""(GPU box) In-kernel timeline of the reference-precision level fusion (csrc/level_fuse_hl.hip) from s_memtime stamps, and its launch time.
Build: make -C slotvps_amd/csrc stampk4hl; run with SLOTVPS_LIB=slotvps_amd/libslotvps_hip_stampk4hl.so (timing alone: any library).
Points of a tile (waves 0 and 4): 0 top, 1 past barrier 1, 2 loads of tile + 2 issued, 3 MFMA chain done, 4 blend + split + out tiles written,
5 past barrier 2, 6 operand tile of tile + 1 committed, 7 out tiles stored.
    python tools/k4hl_stamps.py [--T 40] [--H 256] [--W 512] [--planes 1] [--f32 0] [--taps 1]"""
import argparse, ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd import ops, _lib
ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=40)
ap.add_argument("--H", type=int, default=256)
ap.add_argument("--W", type=int, default=512)
ap.add_argument("--planes", type=int, default=1)
ap.add_argument("--f32", type=int, default=0)
ap.add_argument("--taps", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
cur = torch.randn((a.T, 128, a.H, a.W), generator=g, device=dev)
gprev = torch.randn((a.T, (a.H // 2) * (a.W // 2), 256), generator=g, device=dev) if a.taps else None
w = ops.split_hl(torch.randn((256, 128), generator=g, device=dev) * 0.05)
bias = torch.randn(256, generator=g, device=dev)
fn = lambda: ops.level_fuse_hl_g(cur, gprev, w, bias, a.H, a.W, planes=bool(a.planes), f32=bool(a.f32))
for _ in range(20):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    fn()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 10 * 1e3
px = a.T * a.H * a.W
byts = px * (512 + (256 if a.taps else 0) + 1024 * (a.planes + a.f32))
print(f"level_fuse_hl taps={a.taps} planes={a.planes} f32={a.f32}: {us:9.1f} us  {byts / us / 1e3:7.0f} GB/s algorithmic", flush=True)
lib = _lib.load()
if hasattr(lib, "svps_k4hl_debug_read"):
    buf = (ctypes.c_ulonglong * 128)()
    lib.svps_k4hl_debug_read.restype = ctypes.c_int
    assert lib.svps_k4hl_debug_read(buf) == 0
    st = np.array(list(buf), dtype=np.int64).reshape(2, 8, 8)
    for role, name in ((0, "wave 0"), (1, "wave 4")):
        for it in range(1, 7):
            d = st[role, it] - st[role, it, 0]
            nxt = st[role, it + 1, 0] - st[role, it, 0]
            print(f"{name} tile {it + 8}: " + " ".join(f"p{k}={int(d[k]):6d}" for k in range(8)) + f"  | next top {int(nxt)}")
