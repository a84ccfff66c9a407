"""Device time of K9 (svps_bgemm) on the shapes the slot side uses it for, against torch.matmul (the GEMM library in fp32)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(s, generator=g, device=dev)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


G, L, C = 32, 500, 256
q, k, v = rnd(G, L, C), rnd(G, L, C), rnd(G, L, C)
at = torch.softmax(rnd(G, L, L), -1)
T, LP = 160, 128
q2, ytab, xtab, a1 = rnd(T, LP, 256), rnd(256, 128), rnd(512, 128), rnd(T, LP)
x, wc, bc = rnd(16000, 256), rnd(20, 256), rnd(20)
cases = [
    ("temporal logits k q^T  [32 x 500 x 500 x 256]", lambda: ops.bgemm(k, q), lambda: k @ q.transpose(1, 2)),
    ("temporal out attn^T v  [32 x 500 x 256 x 500]", lambda: ops.bgemm(at.transpose(1, 2), v.transpose(1, 2)), lambda: at.transpose(1, 2) @ v),
    ("cy [160 x 256 x 128 x 128] + bias", lambda: ops.bgemm(ytab, q2[:, :, :128], bias=a1), lambda: torch.baddbmm(a1[:, None, :], ytab.expand(T, -1, -1), q2[:, :, :128].transpose(1, 2))),
    ("cx [160 x 512 x 128 x 128]", lambda: ops.bgemm(xtab, q2[:, :, 128:]), lambda: torch.matmul(xtab, q2[:, :, 128:].transpose(1, 2))),
    ("class projection [16000 x 20 x 256]", lambda: ops.bgemm(x, wc, bias=bc), lambda: torch.nn.functional.linear(x, wc, bc)),
]
for name, f1, f2 in cases:
    print(f"{name:48s}: K9 {timeit(f1):7.1f} us   library fp32 {timeit(f2):7.1f} us", flush=True)
