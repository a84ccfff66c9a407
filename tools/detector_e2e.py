"""Informational: whole-detector timing of one synthetic clip at a config's own geometry (PyTorch trunk + HIP head + GPU
post-process + tracker), one JSON line. Random weights, a fixed slot -> class table so that segments survive the post-process.
    python tools/detector_e2e.py [--config configs/swinL_fpn_slotvps_mi355x.py] [--frames N] [--profile-tower 1]"""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slotvps_amd.config import Config
from slotvps_amd.registry import build_detector
from slotvps_amd.parallel import size_host_pools
if os.environ.get('SVPS_SIZE_POOLS', '1') == '1':
    size_host_pools()

ap = argparse.ArgumentParser()
ap.add_argument("--config", default=os.path.join(ROOT, "configs", "r50_fpn_slotvps_mi355x.py"))
ap.add_argument("--frames", type=int, default=0)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--graph", type=int, default=1)
ap.add_argument("--profile-tower", type=int, default=0)
ap.add_argument("--mode", "--map-dtype", dest="map_dtype", default=None,
                help="head mode (MultiScaleDynamicMaskHead.MODES): fp16x2, fp32, bf16, fp16; default: the config's own (fp16x2)")
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = Config.fromfile(a.config)
torch.manual_seed(0)
det = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg).to(dev).eval()
det.use_graph = bool(a.graph)
if a.map_dtype is not None:
    det.image_model.dynamic_mask_head.set_mode(a.map_dtype)
a.map_dtype = det.image_model.dynamic_mask_head.mode
T = a.frames or cfg.clip["frames"]
H, W = cfg.clip["height"], cfg.clip["width"]
L = det.image_model.init_mask_query.weight.shape[0]
nc = det.num_classes
div = 100000 if nc in (23, 24) else 10000
imgs = torch.randn(T, 3, H, W, device=dev)
table = torch.zeros(L, nc, device=dev)
table[torch.arange(L), torch.arange(L) % (nc - 1)] = 12.0
with torch.no_grad():
    det.image_model.fg_bn.weight.fill_(40.0)
base = det.head_path
det.head_path = lambda f: (lambda lg, em, mk: (lg + table, em, mk))(*base(f))
metas = [dict(iid=div + 1 + t, ori_shape=(H, W, 3), img_shape=(H, W, 3), filename=f"f{t}.png") for t in range(T)]


def timed(fn, n=a.iters):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


with torch.no_grad():
    t_trunk, (feats, fcn) = timed(lambda: det.trunk(imgs))
    t_head, _ = timed(lambda: det.head_path(feats))
    t_all, res = timed(lambda: det.clip_test(imgs, metas))
    # the post-process + tracker timed DIRECTLY (device idle at its start and end) next to the figure by subtraction, which carries the
    # run-to-run noise of three separately timed parts
    direct = []
    orig_cr = det._clip_results
    def timed_cr(*args, **kw):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = orig_cr(*args, **kw)
        torch.cuda.synchronize(); direct.append((time.perf_counter() - t0) * 1e3)
        return out
    det._clip_results = timed_cr
    for _ in range(2 + a.iters):
        det.clip_test(imgs, metas)
    det._clip_results = orig_cr
    t_post = sum(direct[2:]) / max(1, len(direct) - 2)
    im = det.image_model
    t_bb, x = timed(lambda: im.backbone(imgs))
    t_neck, xn = timed(lambda: im.neck(x))
    t_ups, _ = timed(lambda: det.extract_semantic_feats(xn))
    det.trunk_bf16 = True
    t_all16, _ = timed(lambda: det.clip_test(imgs, metas))
    det.trunk_bf16 = False
print(json.dumps({"config": os.path.basename(a.config), "map_dtype": a.map_dtype, "backbone": type(im.backbone).__name__, "clip": [T, H, W], "slots": L,
                  "frames_per_s": round(T / t_all * 1e3, 2), "ms_per_clip": round(t_all, 1), "trunk_ms": round(t_trunk, 1),
                  "backbone_ms": round(t_bb, 1), "fpn_ms": round(t_neck, 1), "semantic_tower_ms": round(t_ups, 1),
                  "slot_head_ms": round(t_head, 1), "post_process_and_tracker_ms": round(t_post, 1),
                  "post_process_and_tracker_ms_by_subtraction": round(t_all - t_trunk - t_head, 1),
                  "frames_per_s_bf16_trunk": round(T / t_all16 * 1e3, 2),
                  "segments_per_frame": [len(r["panoptic_cls_inds"]) for r in res],
                  "peak_mem_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1)}), flush=True)
if a.profile_tower:
    from torch.profiler import profile, ProfilerActivity
    with torch.no_grad(), profile(activities=[ProfilerActivity.CUDA]) as prof:
        det.extract_semantic_feats(xn); torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=70))
