"""(GPU box) In-kernel timeline of the reference-precision statistics kernel (csrc/retr_stats_hl.hip) from s_memtime stamps.
Build: make -C slotvps_amd/csrc stampshl; run with SLOTVPS_LIB=slotvps_amd/libslotvps_hip_stampshl.so.
Points of a tile (key / value wave of quarter 0): 0 top, key: 3 chain + sums done, 4 past barrier 1, 5 tables of tile + 1 summed, 6 loads of
tile + 2 issued, 7 tile + 1 staged; value: 1 start values, 2 loads issued, 3 finish done, 4 past barrier 1, 5 chain + sums done, 7 staged."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd import ops, _lib
from slotvps_amd.slot_head import MaskDynamicConv
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = MaskDynamicConv(256).to(dev).eval()
T, H, W = 40, 256, 512
f = ops.split_hl(torch.randn((T, H * W, 256), device=dev))
tabs = ops.pos_embed_sine_tables(H, W, 256, dev)
c = m._fused_consts()
tyk, txk, rbv, tiled = m.stats_hl_tables(tabs)
for _ in range(30):
    ops.retr_stats_hl(f, H, W, tyk, txk, c["rk"], c["rk_lo"], 1e-5, c["rv"], c["rv_lo"], rbv, 1e-5, tx_tiled=tiled)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 128)()
lib.svps_shl_debug_read.restype = ctypes.c_int
assert lib.svps_shl_debug_read(buf) == 0
st = np.array(list(buf), dtype=np.int64).reshape(2, 8, 8)
for role, name in ((0, "key  "), (1, "value")):
    for it in range(1, 7):
        d = st[role, it] - st[role, it, 0]
        nxt = st[role, it + 1, 0] - st[role, it, 0]
        print(f"{name} tile {it + 8}: " + " ".join(f"p{k}={int(d[k]):6d}" for k in range(8) if st[role, it, k]) + f"  | next top {int(nxt)}")
print("key top - value top per tile:", [int(st[0, it, 0] - st[1, it, 0]) for it in range(8)])
