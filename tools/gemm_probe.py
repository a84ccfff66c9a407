"""fp32 vs split-bf16 vs `allow_tf32` on the slot-side GEMM shapes (16 stacked clips: 8000 rows): time (HIP events) and error.

MI355X has no TF32 matrix instruction; with allow_tf32 hipBLASLt runs an fp32 GEMM as split-bf16 products."""
import torch
dev = torch.device("cuda:0")


def timeit(fn, n=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def split(x):
    hi = x.to(torch.bfloat16)
    return hi, (x - hi.float()).to(torch.bfloat16)


for (M, K, N) in [(8000, 256, 2048), (8000, 2048, 256), (8000, 256, 256), (8000, 256, 768), (8000, 256, 1024), (8000, 1024, 256)]:
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) * 0.05
    b = torch.randn(N, device=dev, generator=g)
    ref = x.double() @ w.double().t() + b.double()
    sc = ref.abs().max().item()
    torch.backends.cuda.matmul.allow_tf32 = False
    y32 = torch.addmm(b, x, w.t())
    t32 = timeit(lambda: torch.addmm(b, x, w.t()))
    torch.backends.cuda.matmul.allow_tf32 = True
    ytf = torch.addmm(b, x, w.t())
    ttf = timeit(lambda: torch.addmm(b, x, w.t()))
    torch.backends.cuda.matmul.allow_tf32 = False
    xh, xl = split(x)
    wh, wl = split(w)
    xs, ws = torch.cat([xh, xl, xh], 1).contiguous(), torch.cat([wh, wh, wl], 1).contiguous()
    y3 = torch.mm(xs, ws.t(), out_dtype=torch.float32) + b
    t3 = timeit(lambda: torch.mm(xs, ws.t(), out_dtype=torch.float32))
    err = lambda y: (y.double() - ref).abs().max().item() / sc
    print(f"M{M} K{K} N{N}: fp32 {t32:6.1f} us err {err(y32):.1e} | allow_tf32 {ttf:6.1f} us err {err(ytf):.1e} | "
          f"split-bf16 [hi,lo,hi]x[hi,hi,lo] {t3:6.1f} us err {err(y3):.1e} (+ operand split)")
