"""(GPU box) Where the post-process + tracker time of detector.clip_test goes: the phases of VPS_Temporal_Slots._clip_results timed one by
one (host wall time with a device synchronisation after each), on the synthetic R50 clip of tools/detector_e2e.py."""
import json, os, sys, time
import numpy as np
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slotvps_amd.config import Config
from slotvps_amd.registry import build_detector
from slotvps_amd.parallel import size_host_pools
if os.environ.get('SVPS_SIZE_POOLS', '1') == '1':
    size_host_pools()

dev = torch.device("cuda:0")
cfg = Config.fromfile(os.path.join(ROOT, "configs", (sys.argv[1] if len(sys.argv) > 1 else "r50_fpn_slotvps_mi355x") + ".py"))
torch.manual_seed(0)
det = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg).to(dev).eval()
det.use_graph = True
T, H, W = cfg.clip["frames"], cfg.clip["height"], cfg.clip["width"]
L, nc = det.image_model.init_mask_query.weight.shape[0], det.num_classes
imgs = torch.randn(T, 3, H, W, device=dev)
table = torch.zeros(L, nc, device=dev)
table[torch.arange(L), torch.arange(L) % (nc - 1)] = 12.0
with torch.no_grad():
    det.image_model.fg_bn.weight.fill_(40.0)
    feats, fcn = det.trunk(imgs)
    pp = det.postprocess_panoptic

    def once():
        t = {}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        logits, embeds, masks = det.head_path(feats)
        logits = logits + table
        torch.cuda.synchronize(); t["head"] = time.perf_counter() - t0; t0 = time.perf_counter()
        results = pp.forward_clip(logits, masks, (H, W))
        torch.cuda.synchronize(); t["forward_clip"] = time.perf_counter() - t0; t0 = time.perf_counter()
        seg = [embeds[i][r.slot_index] for i, r in enumerate(results)]
        emb_h = det.temporal_track_head._embed(torch.cat(seg)).cpu().numpy()
        t["tracker_embed_copy"] = time.perf_counter() - t0; t0 = time.perf_counter()
        pans = pp.panoptic_ids_clip(results, det.stuff_num)
        torch.cuda.synchronize(); t["panoptic_ids_clip"] = time.perf_counter() - t0; t0 = time.perf_counter()
        f = fcn if fcn.shape[-2:] == (H, W) else F.interpolate(fcn, size=(H, W), mode="bilinear", align_corners=False)
        ids = f.argmax(dim=1)
        torch.cuda.synchronize(); t["fcn_argmax"] = time.perf_counter() - t0
        return t
    for _ in range(3):
        once()
    acc = {}
    for _ in range(10):
        for k, v in once().items():
            acc[k] = acc.get(k, 0.0) + v
print(json.dumps({k: round(v / 10 * 1e3, 3) for k, v in acc.items()} | {"fcn_shape": list(fcn.shape), "config": cfg.filename if hasattr(cfg, "filename") else ""}))
