"""Phase timeline of K3 from s_memtime stamps (variant library built with -DSVPS_K3_STAMP)."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd import ops, _lib, synth
dev = torch.device("cuda:0")
p = synth.make_params(synth.retriever_shapes(""), 1)
P = {k: torch.from_numpy(v).to(dev) for k, v in p.items()}
wk, wv = P["to_k.weight"].to(torch.bfloat16).contiguous(), P["to_v.weight"].to(torch.bfloat16).contiguous()
H, W, T = 256, 512, 5
f = torch.randn((T, H * W, 256), device=dev).to(torch.bfloat16)
tabs = ops.pos_embed_sine_tables(H, W, 256, dev)
fn = lambda: ops.kv_project(f, H, W, tabs, wk, P["to_k.bias"], P["norm_k.weight"], P["norm_k.bias"], 1e-5, wv,
                            P["to_v.bias"], P["norm_v.weight"], P["norm_v.bias"], 1e-5)
t0 = time.time()
while time.time() - t0 < 0.3:
    for _ in range(10): fn()
    torch.cuda.synchronize()
fn(); torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros((8, 8, 16), dtype=np.uint64)
rc = lib.svps_k3_debug_read(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0
names = {0: "pre-B1", 1: "B1 done", 2: "dma issued", 3: "stores issued", 4: "mfma done", 5: "stats done", 6: "heavy ret",
         7: "pre-B2", 8: "B2 done", 9: "light done", 10: "end", 11: "light(v) done"}
base = buf[:, 0, 0].min()
for w in (0, 4):
    print(f"--- wave {w} ({'key' if w < 4 else 'value'}) ; s_memtime ticks (shader clock cycles: about 2 GHz, checked against the kernel time) relative")
    for it in range(1, 5):
        row = buf[w, it].astype(np.int64) - int(base)
        print(f"it {it+8}: " + "  ".join(f"{names[i]}={row[i]}" for i in range(12) if buf[w, it, i]))
per_tile = (buf[0, 7, 0].astype(np.int64) - buf[0, 1, 0].astype(np.int64)) / 6
print("ticks per tile", per_tile)
