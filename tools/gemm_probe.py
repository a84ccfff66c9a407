"""K8 (split-bf16 slot GEMM) against the GEMM library in fp32 on the slot-side shapes: device time (HIP events) and error."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)


def t(fn, n=20):
    """Device time per call: n calls captured into one hipGraph (no host launch overhead between them), replayed 5 times."""
    for _ in range(3):
        o = fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(n):
                o = fn()
    torch.cuda.current_stream().wait_stream(side)
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3, o


tot_a = tot_b = 0.0
for (M, K, N, act, cnt) in [(8000, 256, 768, None, 7), (8000, 256, 256, None, 7 * 7), (10240, 256, 256, None, 7), (8000, 272, 256, None, 7),
                            (8000, 256, 2048, "gelu", 7), (8000, 2048, 256, None, 7), (8000, 256, 1024, "relu", 4), (8000, 1024, 256, None, 4)]:
    x = torch.randn((M, K), generator=g, device=dev)
    w = torch.randn((N, K), generator=g, device=dev) / K ** 0.5
    b = torch.randn((N,), generator=g, device=dev)
    wp = ops.pack_b_fragments(w)
    code = {None: 0, "relu": 1, "gelu": 2}[act]
    ta, ya = t(lambda: ops.slot_gemm(x, wp, b, code))
    f = (lambda y: y) if act is None else (F.relu if act == "relu" else F.gelu)
    tb, yb = t(lambda: f(F.linear(x, w, b)))
    ref = f(F.linear(x.double(), w.double(), b.double()))
    tot_a += ta * cnt
    tot_b += tb * cnt
    print(f"M={M} K={K} N={N} act={act}: K8 {ta:6.1f} us  library fp32 {tb:6.1f} us   err K8 {(ya - ref).abs().max().item():.1e} lib {(yb - ref).abs().max().item():.1e}")
print(f"per step (7 stages, counts as in the head): K8 {tot_a / 1e3:.2f} ms, library {tot_b / 1e3:.2f} ms")
