"""(GPU box) K7' (deform_conv_fused) alone at the semantic tower's shapes, warmed clocks. SLOTVPS_LIB selects an ablation build."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd.dcn import deform_conv_fused_pm, pack_weight_fragments
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
shapes = [(5, 256 >> i, 512 >> i, c, o) for i in range(4) for (c, o) in ((256, 256), (256, 128), (128, 128))]
if os.environ.get("VIPER"):
    shapes = [(10, 272 >> i, 480 >> i, c, o) for i in range(4) for (c, o) in ((256, 256), (256, 128), (128, 128))]
for (N, H, W, C, O) in shapes:
    x = torch.randn((N, H, W, C), generator=g, device=dev)
    off = 0.7 * torch.randn((N, 18, H, W), generator=g, device=dev)
    wp = pack_weight_fragments(torch.randn((O, C, 3, 3), generator=g, device=dev) / (3 * C ** 0.5))
    t0 = time.time()
    while time.time() - t0 < 0.4:
        deform_conv_fused_pm(x, off, wp, O, 1, 1, 1); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        deform_conv_fused_pm(x, off, wp, O, 1, 1, 1)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    fl = N * H * W * O * 9 * C * 2 * 3
    print(f"tile {os.environ.get('SVPS_K7_TILE', 'rule'):4s} N={N} {H}x{W} C={C} O={O}: {us:8.1f} us, {fl / us / 1e9:.3f} PFLOP/s executed")
