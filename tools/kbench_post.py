"""Timing of the GPU panoptic post-process at BASELINE size (one 1024x2048 frame from 256x512 logits)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd import synth, ops, _lib
from slotvps_amd.postprocess import PostProcessPanopticInstances
dev = torch.device("cuda:0")
for nk in (12, 30, 60):
    logits, masks = synth.make_post_case(21, 100, 256, 512, 20, nk)
    pp = PostProcessPanopticInstances(is_thing_map={i: i > 10 for i in range(20)}, apply_mask_removal=True,
                                      apply_mask_removal_only_ins=True)
    tl, tm = torch.from_numpy(logits).to(dev), torch.from_numpy(masks).to(dev)
    for _ in range(3):
        res = pp.forward_tensors(tl, tm, (1024, 2048)); ids = pp.panoptic_ids(res)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with ops.KernelTimer() as kt:
        for _ in range(10):
            res = pp.forward_tensors(tl, tm, (1024, 2048)); ids = pp.panoptic_ids(res)
        torch.cuda.synchronize()
        ms, n = kt.collect(_lib.KERNEL_PANOPTIC_POST)
    wall = (time.perf_counter() - t0) / 10 * 1e3
    print(f"selected {nk}: kept {len(res.labels)}  wall {wall:.2f} ms/frame, K6 kernels {ms/10:.3f} ms/frame in {n//10} launches")
