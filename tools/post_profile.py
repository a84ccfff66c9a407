"""Where the per-frame post-process + tracker time goes (host wall clock, one synthetic 1024x2048 frame)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd import synth
from slotvps_amd.postprocess import PostProcessPanopticInstances
dev = torch.device("cuda:0")
logits, masks = synth.make_post_case(21, 100, 256, 512, 20, 30)
tl, tm = torch.from_numpy(logits).to(dev), torch.from_numpy(masks).to(dev)
pp = PostProcessPanopticInstances(is_thing_map={i: i > 10 for i in range(20)}, threshold=0.85, fraction_threshold=0.03,
                                  pixel_threshold=0.4, apply_mask_removal=True, apply_mask_removal_only_ins=True)
def wall(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, out
t1, res = wall(lambda: pp.forward_tensors(tl, tm, (1024, 2048)))
t2, _ = wall(lambda: pp.panoptic_ids(res))
fcn = torch.randn(1, 19, 1024, 2048, device=dev)
t3, _ = wall(lambda: fcn.argmax(dim=1))
print(f"forward_tensors {t1:.2f} ms, panoptic_ids {t2:.2f} ms, semantic argmax {t3:.2f} ms  (kept {len(res.area)})")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    r = pp.forward_tensors(tl, tm, (1024, 2048)); pp.panoptic_ids(r)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
