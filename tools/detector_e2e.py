"""Informational: whole-detector timing on one synthetic 1024x2048 T=5 clip (PyTorch trunk + HIP head + GPU post-process)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd.config import Config
from slotvps_amd.registry import build_detector
dev = torch.device("cuda:0")
cfg = Config.fromfile(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "r50_fpn_slotvps_mi355x.py"))
torch.manual_seed(0)
det = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg).to(dev).eval()
T, H, W = 5, 1024, 2048
imgs = torch.randn(T, 3, H, W, device=dev)
table = torch.zeros(100, 20, device=dev); table[torch.arange(100), torch.arange(100) % 19] = 12.0
with torch.no_grad():
    det.image_model.fg_bn.weight.fill_(40.0)
base = det.head_path
det.head_path = lambda f: (lambda lg, em, mk: (lg + table, em, mk))(*base(f))
metas = [dict(iid=10001 + t, ori_shape=(H, W, 3), img_shape=(H, W, 3), filename=f"f{t}.png") for t in range(T)]
def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out
with torch.no_grad():
    t_trunk, (feats, fcn) = timed(lambda: det.trunk(imgs))
    t_head, (lg, em, mk) = timed(lambda: det.head_path(feats))
    t_all, res = timed(lambda: det.clip_test(imgs, metas))
print(f"trunk (R50 + FPN + UPSNetFPN/K7 + conv_trans, PyTorch): {t_trunk:.1f} ms per clip ({t_trunk / T:.1f} ms/frame)")
print(f"slot head + decode (HIP, eager, one clip): {t_head:.1f} ms per clip")
print(f"clip_test total incl. post-process + tracker: {t_all:.1f} ms per clip -> {T / t_all * 1e3:.1f} frames/s; post-process+tracker ~ {t_all - t_trunk - t_head:.1f} ms")
print("segments per frame:", [len(r["panoptic_cls_inds"]) for r in res])
with torch.no_grad():
    im = det.image_model
    t_bb, x = timed(lambda: im.backbone(imgs))
    t_neck, xn = timed(lambda: im.neck(x))
    t_ups, _ = timed(lambda: det.extract_semantic_feats(xn))
print(f"backbone {t_bb:.1f} ms, FPN {t_neck:.1f} ms, UPSNetFPN {t_ups:.1f} ms per T=5 clip")
det.trunk_bf16 = True
with torch.no_grad():
    t_trunk16, _ = timed(lambda: det.trunk(imgs))
    t_all16, _ = timed(lambda: det.clip_test(imgs, metas))
print(f"with trunk_bf16 (autocast): trunk {t_trunk16:.1f} ms, clip_test {t_all16:.1f} ms per clip -> {T / t_all16 * 1e3:.1f} frames/s")
det.trunk_bf16 = False
from torch.profiler import profile, ProfilerActivity
with torch.no_grad(), profile(activities=[ProfilerActivity.CUDA]) as prof:
    det.extract_semantic_feats(xn); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=70))
