"""In-kernel timeline of K2's fast path (argmax-only mode by default) from s_memtime stamps. Needs the diagnostic library:
    make -C slotvps_amd/csrc stampk2
    SLOTVPS_LIB=slotvps_amd/libslotvps_hip_stampk2.so python tools/k2_stamps.py [--T 40] [--logits]
Points of an iteration (all four waves of one workgroup): 0 top, 1 own DMA pieces of the tile landed, 2 past the barrier, 3 DMA of
tile it + 3 requested, 4 (wave 0: cross-wave argmax of tile it - 2 stored), 5 the 32 MFMAs issued with the epilogue of tile it - 1
between them, 6 candidates written (logits mode: row segments stored), 7 accumulators of the chain read back."""
import argparse, ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd import ops, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=40)
ap.add_argument("--H", type=int, default=256)
ap.add_argument("--W", type=int, default=512)
ap.add_argument("--logits", action="store_true")
ap.add_argument("--hl", action="store_true", help="the fp16x2 form (hi / lo planes, logits + argmax)")
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
f = torch.randn((a.T, a.H * a.W, 256), generator=g, device=dev).to(torch.bfloat16)
e = torch.relu(torch.randn((a.T, 100, 256), generator=g, device=dev))
one, zero = torch.ones(256, device=dev), torch.zeros(256, device=dev)
if a.hl:
    planes = ops.split_hl(2.0 * torch.randn((a.T, a.H * a.W, 256), generator=g, device=dev))
    run = lambda: ops.mask_decode_hl(planes, e, one, zero, 0.1, 0.0, want_argmax=True)
else:
    run = lambda: ops.mask_decode(f, e, one, zero, 0.1, 0.0, want_argmax=True, want_logits=a.logits)
t0 = time.time()
while time.time() - t0 < 1.5:
    for _ in range(5):
        run()
    torch.cuda.synchronize()
lib = _lib.load()
st = np.zeros((4, 8, 8), dtype=np.uint64)
lib.svps_k2_debug_read.argtypes = [ctypes.c_void_p]
assert lib.svps_k2_debug_read(st.ctypes.data_as(ctypes.c_void_p)) == 0
st = st.astype(np.int64)
base = st[:, 0, 0].min()
for w in range(4):
    print(f"wave {w}")
    for i in range(8):
        row = st[w, i]
        nxt = st[w, i + 1][0] if i + 1 < 8 else None
        d = [int(row[k + 1] - row[k]) for k in range(7)]
        print(f"  it {8 + i}: top at {int(row[0] - base):6d}; deltas {d}" + (f"; iteration {int(nxt - row[0])}" if nxt else ""))
