"""(GPU box) K7' (deform_conv_fused) alone at the semantic tower's shapes, warmed clocks. SLOTVPS_LIB selects an ablation build."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd.dcn import deform_conv_fused_pm, pack_weight_fragments
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for (N, H, W, C, O) in ((5, 256, 512, 256, 256), (5, 256, 512, 256, 128), (5, 256, 512, 128, 128), (5, 64, 128, 256, 256)):
    x = torch.randn((N, H, W, C), generator=g, device=dev)
    off = 0.7 * torch.randn((N, 18, H, W), generator=g, device=dev)
    wp = pack_weight_fragments(torch.randn((O, C, 3, 3), generator=g, device=dev) / (3 * C ** 0.5))
    t0 = time.time()
    while time.time() - t0 < 1.0:
        deform_conv_fused_pm(x, off, wp, O, 1, 1, 1); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        deform_conv_fused_pm(x, off, wp, O, 1, 1, 1)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    fl = N * H * W * O * 9 * C * 2 * 3
    print(f"{os.environ.get('SLOTVPS_LIB', 'product')[-22:]:22s} N={N} {H}x{W} C={C} O={O}: {us:8.1f} us, {fl / us / 1e9:.3f} PFLOP/s executed")
