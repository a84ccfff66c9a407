"""In-kernel timeline of K4 v4 (level fusion, wave-specialised) from s_memtime stamps. Needs the diagnostic library:
    make -C slotvps_amd/csrc stampk4
    SLOTVPS_LIB=slotvps_amd/libslotvps_hip_stampk4.so python tools/k4_stamps.py [--T 40]
Matrix wave 0: 0 before the barrier, 1 after it, 2 after the 48 MFMAs (then the out-tile write).
Helper wave 0: 0 before the barrier, 1 after it, 2 tap requests issued, 3 map loads issued, 4 out-tile stores issued, 5 map loads of
tile it+1 landed, 6 operand tile it+1 built, 7 tap requests of tile it+2 landed."""
import argparse, ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd import ops, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=40)
ap.add_argument("--H", type=int, default=256)
ap.add_argument("--W", type=int, default=512)
ap.add_argument("--rows", type=int, default=0, help="1: the default step's form since round 4 - the incoming map as bf16 pixel-major rows, "
                                                     "the bf16 storage policy in the fp16 encoding (bf16_values)")
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
cur = torch.randn((a.T, 128, a.H, a.W), generator=g, device=dev)
prev = torch.randn((a.T, a.H * a.W // 4, 256), generator=g, device=dev).to(torch.bfloat16)
wc = (torch.randn((256, 384), generator=g, device=dev) * 0.05).to(torch.bfloat16)
bc = torch.zeros(256, device=dev)
kw = {}
if a.rows:
    cur = cur.reshape(a.T, 128, -1).transpose(1, 2).to(torch.bfloat16).contiguous()
    prev = prev.float().to(torch.float16)
    kw = dict(bf16_values=True)
t0 = time.time()
while time.time() - t0 < 1.5:
    for _ in range(5):
        ops.level_fuse(cur, prev, wc, bc, a.H, a.W, **kw)
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.level_fuse(cur, prev, wc, bc, a.H, a.W, **kw)
e1.record()
torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1) * 100:.1f} us per launch ({os.environ.get('SLOTVPS_LIB', 'product library')}), T = {a.T}, rows = {a.rows}")
lib = _lib.load()
if not hasattr(lib, "svps_k4_debug_read"):
    sys.exit(0)                                   # a library without stamps: the timing line above is all there is
st = np.zeros((2, 8, 8), dtype=np.uint64)
lib.svps_k4_debug_read.argtypes = [ctypes.c_void_p]
assert lib.svps_k4_debug_read(st.ctypes.data_as(ctypes.c_void_p)) == 0
st = st.astype(np.int64)
for role, name, npts in ((0, "matrix wave 0", 3), (1, "helper wave 0", 8)):
    print(name)
    for i in range(8):
        row = st[role, i]
        if row[0] == 0:
            continue
        nxt = st[role, i + 1][0] if i + 1 < 8 and st[role, i + 1][0] else None
        d = [int(row[k + 1] - row[k]) for k in range(npts - 1)]
        print(f"  it {8 + i}: deltas between points {d}" + (f", to the next iteration's point 0: {int(nxt - row[npts - 1])}; iteration {int(nxt - row[0])}" if nxt else ""))
