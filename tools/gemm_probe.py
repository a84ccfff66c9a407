import os, time, torch
dev = torch.device("cuda:0")
def timeit(fn, n=50):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e6
def split(x):
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return hi, lo
for (M, K, N) in [(8000, 256, 2048), (8000, 2048, 256), (8000, 256, 256), (8000, 256, 768), (8000, 1024, 256)]:
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) * 0.05
    b = torch.randn(N, device=dev, generator=g)
    ref = (x.double() @ w.double().t() + b.double())
    y32 = torch.addmm(b, x, w.t())
    t32 = timeit(lambda: torch.addmm(b, x, w.t()))
    xh, xl = split(x); wh, wl = split(w)
    xs = torch.cat([xh, xl, xh], 1).contiguous(); ws = torch.cat([wh, wh, wl], 1).contiguous()
    y3 = torch.mm(xs, ws.t(), out_dtype=torch.float32) + b
    t3 = timeit(lambda: torch.mm(xs, ws.t(), out_dtype=torch.float32))
    try:
        ya = torch.addmm(b, xs, ws.t(), out_dtype=torch.float32)
        ta = timeit(lambda: torch.addmm(b, xs, ws.t(), out_dtype=torch.float32))
        ea = (ya.double() - ref).abs().max().item()
    except Exception as e:
        ta, ea = -1, repr(e)[:80]
    t1 = timeit(lambda: torch.mm(xh, wh.t(), out_dtype=torch.float32))
    tsplit = timeit(lambda: torch.cat([*split(x), x.to(torch.bfloat16)], 1))
    sc = ref.abs().max().item()
    print(f"M{M} K{K} N{N}: fp32 {t32:.1f} us err {(y32.double()-ref).abs().max().item()/sc:.2e} | bf16x3 {t3:.1f} us err {(y3.double()-ref).abs().max().item()/sc:.2e} | addmm-out {ta:.1f} {ea} | bf16x1 {t1:.1f} us | torch split {tsplit:.1f} us")
torch.backends.cuda.matmul.allow_tf32 = True
x = torch.randn(8000, 256, device=dev); w = torch.randn(2048, 256, device=dev) * 0.05
ref = x.double() @ w.double().t()
y = x @ w.t()
print("allow_tf32:", timeit(lambda: x @ w.t()), "us err", ((y.double() - ref).abs().max() / ref.abs().max()).item())
