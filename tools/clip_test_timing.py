import json, os, sys, time
import torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from slotvps_amd.config import Config
from slotvps_amd.registry import build_detector
from slotvps_amd.parallel import size_host_pools
if os.environ.get('SVPS_SIZE_POOLS', '1') == '1':
    size_host_pools()
dev = torch.device("cuda:0")
cfg = Config.fromfile(os.path.join(ROOT, "configs", "r50_fpn_slotvps_mi355x.py"))
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
base = det.head_path
det.head_path = lambda f: (lambda lg, em, mk: (lg + table, em, mk))(*base(f))
metas = [dict(iid=10001 + t, ori_shape=(H, W, 3), img_shape=(H, W, 3), filename=f"f{t}.png") for t in range(T)]
import slotvps_amd.postprocess as P
orig = P.PostProcessPanopticInstances._clip_on_device
marks = {}
def wrapped(self, *a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = orig(self, *a, **k)
    torch.cuda.synchronize(); marks.setdefault("on_device", []).append((time.perf_counter() - t0) * 1e3)
    return r
if os.environ.get("WRAP", "1") == "1":
    P.PostProcessPanopticInstances._clip_on_device = wrapped
import gc
gc_log = []
gc.callbacks.append(lambda phase, info: gc_log.append((phase, info.get("generation"), time.perf_counter())))
sect = {}
def wrap_sync(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); sect.setdefault(name, []).append(round((time.perf_counter() - t0) * 1e3, 1))
        return r
    setattr(obj, name, g)
ev_log = []
def wrap_events(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        h0 = time.perf_counter(); e0.record()
        r = f(*a, **k)
        e1.record(); ev_log.append((name, h0, time.perf_counter(), e0, e1))
        return r
    setattr(obj, name, g)
if os.environ.get("EVENTS", "0") == "1":
    for n in ("trunk", "head_path", "_clip_results"):
        wrap_events(det, n)
    import slotvps_amd.postprocess as P2
    for n in ("forward_clip", "panoptic_ids_clip"):
        wrap_events(det.postprocess_panoptic, n)
if os.environ.get("SECTIONS", "1") == "1":
    for n in ("trunk", "head_path", "_clip_results"):
        wrap_sync(det, n)
with torch.no_grad():
    times, allocs = [], []
    for i in range(14):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        det.clip_test(imgs, metas)
        torch.cuda.synchronize(); times.append(round((time.perf_counter() - t0) * 1e3, 1))
        ms = torch.cuda.memory_stats()
        allocs.append((ms["num_device_alloc"], ms["num_device_free"], ms["num_alloc_retries"]))
    print("clip_test ms:", times)
    if ev_log:
        torch.cuda.synchronize()
        per = {}
        for name, h0, h1, e0, e1 in ev_log:
            per.setdefault(name, []).append((round((h1 - h0) * 1e3, 1), round(e0.elapsed_time(e1), 1)))
        for name, v in per.items():
            print(name, "(host ms, gpu ms):", v)
    print("sections:", sect)
    print("device allocs/frees/retries:", allocs)
    print("gc events:", [(p_, g_) for p_, g_, _ in gc_log if p_ == "start"])
    print("on_device ms:", [round(x, 2) for x in marks.get("on_device", [])])
    feats, fcn = det.trunk(imgs); torch.cuda.synchronize()
    tt = []
    for i in range(5):
        t0 = time.perf_counter(); det.trunk(imgs); torch.cuda.synchronize(); tt.append(round((time.perf_counter() - t0) * 1e3, 1))
    print("trunk ms:", tt)
