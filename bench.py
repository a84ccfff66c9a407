#!/usr/bin/env python3
"""Benchmark of the MI355X-native Slot-VPS hot path (slot-retriever decode loop + mask decode).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic clips per rank: `--clips-per-launch`
(default 16) independent T=5 clips of 1024x2048 stacked along the frame axis -> four levels of 128-channel
FPN feature maps (32x64 ... 256x512, resident in HBM) -> the 7-stage multi-scale slot head (K4, K3, K1
once per level / stage for all frames of the batch; temporal slot attention stays inside each clip) ->
slot->mask decode of all frames (K2). Weights: the R50-FPN Slot-VPS head architecture with seeded synthetic values
(no checkpoints exist, README.md:25 of the reference). Clips are independent, so ranks share nothing
(weak scaling); the per-clip results (uint8 slot-argmax maps + class logits) are gathered to rank 0
over RCCL inside the timed region.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field definitions):
  value            frames/s, whole job            = n_gpus * steps * T / max-over-ranks wall time
  roofline         K1 (slot_attn_partial_*): algorithmic bytes (k and v read once + q + out) of all K1
                   launches of the timed steps / their summed device time from HIP events recorded on
                   the launch stream, against the 8 TB/s HBM3E peak
  cpu_baseline     the CPU oracle (NumPy port of the reference) on the host cores, bounded sample
  whole_detector   informational (N=1 only, outside the timed region, never part of `value`): one clip through the whole
                   detector - PyTorch-ROCm ResNet-50 + FPN + semantic tower, this path, GPU post-process, tracker
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=5, help="T, frames per clip")
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--slots", type=int, default=100)
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--fast-p", action="store_true", help="bf16 P without the hi/lo split inside K1")
    ap.add_argument("--clips-in-flight", type=int, default=1,
                    help="independent clips per step, each replayed on its own HIP stream (a step then covers that many clips)")
    ap.add_argument("--clips-per-launch", type=int, default=16,
                    help="independent clips stacked along the frame axis of every kernel launch (temporal attention stays per clip)")
    ap.add_argument("--cpu-baseline", type=int, default=1, help="0 to skip the CPU oracle leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    ap.add_argument("--whole-detector", type=int, default=1,
                    help="0 to skip the informational whole-detector leg (PyTorch trunk + this path + post-process + tracker)")
    return ap.parse_args()


def cpu_baseline(a):
    """Time the CPU oracle (NumPy restatement of the reference, fp32) on a bounded sample of the same
    workload: single frames (T=1) of the 1024x2048 head + decode, repeated until ~cpu_seconds."""
    import numpy as np
    from oracle import slotvps_oracle as orc
    from slotvps_amd import synth
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    params = synth.make_params(synth.head_shapes(), 0)
    sizes = synth.level_sizes(a.height, a.width)
    pos = [orc.pos_embed_sine(h, w) for (h, w) in sizes]
    rng = np.random.default_rng(1)
    feats = [[rng.standard_normal((128, h, w)).astype(np.float32) for (h, w) in sizes]]
    slots = synth.make_slots(1, a.slots)
    scale, shift = orc.bn_eval_affine(np.ones(256, np.float32), np.zeros(256, np.float32),
                                      np.zeros(256, np.float32), np.ones(256, np.float32))
    frames, t0 = 0, time.perf_counter()
    while True:
        _, embeds, fused = orc.head_forward(feats, slots, pos, params)
        m = orc.mask_decode(fused[0][3], embeds[0][-1], scale, shift, np.float32(0.1), np.float32(0.0))
        orc.slot_argmax(m)
        frames += 1
        el = time.perf_counter() - t0
        if el >= a.cpu_seconds or frames >= 8:
            break
    return {"value": round(frames / el, 4), "unit": "frames/s", "cores": int(threads), "kind": "port",
            "sample": f"{frames} single-frame clip(s) (T=1) of the same {a.height}x{a.width} L={a.slots} head + "
                      f"mask decode, fp32 NumPy oracle ({threads} BLAS threads for the matrix products; softmax, "
                      f"LayerNorm and resampling are single-threaded NumPy), {el:.1f} s wall"}


def whole_detector_leg(a, dev):
    """Informational, rank 0 at N=1, outside the timed region and never part of `value`: one synthetic T-frame clip through
    the WHOLE detector of configs/r50_fpn_slotvps_mi355x.py - ResNet-50 + FPN + semantic tower in PyTorch-ROCm (fp32, as
    the reference runs them; random weights), the slot head and decode of this library (eager, one clip, no stacking), the
    GPU post-process and the tracker (detector.VPS_Temporal_Slots.clip_test). Says what the hot path is a part of."""
    from slotvps_amd.config import Config
    from slotvps_amd.registry import build_detector
    cfg = Config.fromfile(os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs", "r50_fpn_slotvps_mi355x.py"))
    torch.manual_seed(0)
    det = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg).to(dev).eval()
    T, H, W = a.frames, a.height, a.width
    imgs = torch.randn(T, 3, H, W, device=dev)
    # random-init slots all predict "no object": a fixed slot -> class table lets segments survive the post-process
    table = torch.zeros(a.slots, 20, device=dev)
    table[torch.arange(a.slots), torch.arange(a.slots) % 19] = 12.0
    with torch.no_grad():
        det.image_model.fg_bn.weight.fill_(40.0)
    base = det.head_path
    det.head_path = lambda f: (lambda lg, em, mk: (lg + table, em, mk))(*base(f))
    metas = [dict(iid=10001 + t, ori_shape=(H, W, 3), img_shape=(H, W, 3), filename=f"f{t}.png") for t in range(T)]

    def timed(fn, n=2):
        fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / n * 1e3, out

    with torch.no_grad():
        t_trunk, (feats, _fcn) = timed(lambda: det.trunk(imgs))
        t_head, _ = timed(lambda: det.head_path(feats))
        t_all, res = timed(lambda: det.clip_test(imgs, metas))
        det.trunk_bf16 = True
        t_all16, _ = timed(lambda: det.clip_test(imgs, metas))
    return {"value": round(T / t_all * 1e3, 2), "unit": "frames/s", "ms_per_clip": round(t_all, 2),
            "trunk_ms": round(t_trunk, 2), "slot_head_and_decode_ms": round(t_head, 2),
            "post_process_and_tracker_ms": round(t_all - t_trunk - t_head, 2),
            "value_with_bf16_autocast_trunk": round(T / t_all16 * 1e3, 2),
            "segments_per_frame": [int(len(r["panoptic_cls_inds"])) for r in res],
            "what": f"one {H}x{W} T={T} clip, whole detector, PyTorch fp32 trunk + this library, eager, n_gpus=1; informational"}


def main():
    a = parse()
    from slotvps_amd import _lib, ops, parallel
    from slotvps_amd.clip import SlotClipRunner

    rank, local_rank, world = parallel.init_distributed()
    if world != a.gpus and rank == 0:
        print(f"[bench] note: --gpus {a.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    _lib.load()

    T = a.frames
    cif = max(1, a.clips_in_flight)
    cpl = max(1, a.clips_per_launch)
    n_pool = 2 * cif    # distinct synthetic clips per rank, each resident in its own input slot
    runner = SlotClipRunner(dev, T, a.height, a.width, L=a.slots, param_seed=0, split_p=not a.fast_p,
                            use_graph=not a.no_graph, n_slots=n_pool, clips_per_launch=cpl)
    HWf = runner.sizes[-1][0] * runner.sizes[-1][1]
    # synthetic clips, resident in HBM (in the runner's input slots) before the timed region
    for i in range(n_pool):
        runner.load_clip(runner.random_clip(1234 + rank * 1000 + i), slot=i)
    results = torch.empty((a.steps, cif, cpl * T, HWf), dtype=torch.uint8, device=dev)
    streams = [torch.cuda.Stream(device=dev) for _ in range(cif)] if cif > 1 else None

    def step(i, record):
        if cif == 1:
            out = runner.run(slot=i % n_pool)
            if record:
                results[i, 0].copy_(out["slot_argmax"])
            return
        main_s = torch.cuda.current_stream(dev)
        for j, st in enumerate(streams):                 # independent clips: one hipGraph replay per stream
            st.wait_stream(main_s)
            with torch.cuda.stream(st):
                out = runner.run(slot=(cif * i + j) % n_pool)
                if record:
                    results[i, j].copy_(out["slot_argmax"])
        for st in streams:
            main_s.wait_stream(st)

    for i in range(a.warmup):
        step(i, False)
    if world > 1:
        parallel.gather_to_rank0(results)        # untimed: RCCL sets up its point-to-point connections at first use
    torch.cuda.synchronize(dev)

    # ---------------------------------- timed region -------------------------------------------
    parallel.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i, True)
    gathered = parallel.gather_to_rank0(results)            # RCCL gather of the per-clip results
    torch.cuda.synchronize(dev)
    parallel.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    elapsed = parallel.max_over_ranks(elapsed, dev)
    if rank == 0:
        assert len(gathered) == world and gathered[0].shape == results.shape

    # ------------------- roofline leg: same steps, eager, HIP events around every launch --------------
    roof = None
    if rank == 0:
        eager = runner.use_graph
        runner.use_graph = False
        for i in range(2):
            step(i, False)
        torch.cuda.synchronize(dev)
        with ops.KernelTimer() as kt:
            for i in range(a.steps):
                step(i, False)
            torch.cuda.synchronize(dev)
            k1_ms, k1_n = kt.collect(_lib.KERNEL_SLOT_ATTN)
            fin_ms, fin_n = kt.collect(_lib.KERNEL_SLOT_ATTN_FINISH)
            k2_ms, k2_n = kt.collect(_lib.KERNEL_MASK_DECODE)
            k3_ms, k3_n = kt.collect(_lib.KERNEL_KV_PROJECT)
            k4_ms, k4_n = kt.collect(_lib.KERNEL_LEVEL_FUSE)
        runner.use_graph = eager
        alg = runner.k1_algorithmic_bytes_per_step() * a.steps * cif
        achieved = alg / (k1_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "r01", "k1_pmc_traffic.json")
        wkey = f"{a.height}x{a.width} T={a.frames} L={a.slots} cpl={cpl}"
        rec = None
        if os.path.exists(pmc):
            with open(pmc) as fh:
                rec = json.load(fh)
        if rec is not None and rec.get("workload_key") == wkey:
            traffic = int(rec["traffic_bytes_per_launch"])          # FETCH_SIZE x2 (gfx950) + WRITE_SIZE, avg per K1 launch
            traffic_src = "profiles/r01/k1_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, same workload)"
        # measured on-box ceiling next to the vendor peak: a plain device-to-device copy of 1 GiB (bytes read + written)
        src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        dst = torch.empty_like(src)
        for _ in range(3):
            dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize(dev)
        copy_gbs = 10 * 2 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst
        oth = runner.other_algorithmic_bytes_per_step()
        oth_ms = {"kv_project": k3_ms, "level_fuse": k4_ms, "mask_decode": k2_ms}
        others = {k: {"achieved": round(oth[k] * a.steps * cif / (oth_ms[k] * 1e-3) / 1e9, 1),
                      "frac": round(oth[k] * a.steps * cif / (oth_ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                      "algorithmic_bytes_per_step": int(oth[k])} for k in oth}
        roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "slot_attn_partial_ws", "launches": k1_n, "avg_launch_us": round(k1_ms / k1_n * 1e3, 2),
                "algorithmic_bytes_per_launch_avg": int(alg / k1_n),
                "copy_kernel_ceiling": {"gbps": round(copy_gbs, 1), "frac_of_it": round(achieved / copy_gbs, 4),
                                        "what": "torch device-to-device copy of 1 GiB, bytes read + written"},
                "other_kernels_hbm": others,
                "other_kernels_us_per_clip": {"slot_attn_finish": round(fin_ms / a.steps / cif / cpl * 1e3, 1),
                                              "mask_decode": round(k2_ms / a.steps / cif / cpl * 1e3, 1),
                                              "kv_project": round(k3_ms / a.steps / cif / cpl * 1e3, 1),
                                              "level_fuse": round(k4_ms / a.steps / cif / cpl * 1e3, 1),
                                              "slot_attn_partial": round(k1_ms / a.steps / cif / cpl * 1e3, 1)}}

    if rank == 0:
        frames = world * a.steps * T * cif * cpl
        line = {
            "metric": "frames/sec (whole node), 1024x2048 T=5 clip, R50-FPN Slot-VPS inference",
            "value": round(frames / elapsed, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"slot-retriever decode hot path: R50-FPN Slot-VPS head (7 stages over 4 FPN "
                                   f"levels) + slot->mask decode, {a.height}x{a.width} T={T} clips, {a.slots} slots, "
                                   f"{cpl} independent clips stacked per launch x {cif} in flight per step, "
                                   f"synthetic FPN features resident in HBM; backbone/FPN not in the step",
                       "clip_frames": T, "slots": a.slots, "levels": [list(s) for s in runner.sizes],
                       "parallelism": f"clip-parallel x{world}", "hipgraph": not a.no_graph, "clips_in_flight": cif, "clips_per_launch": cpl,
                       "k1_split_p": not a.fast_p},
            "roofline": roof,
        }
        if a.cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(a)
        if a.whole_detector and world == 1:
            try:
                line["whole_detector"] = whole_detector_leg(a, dev)
            except Exception as e:                       # informational leg: never costs the bench line
                line["whole_detector"] = {"value": None, "error": f"{type(e).__name__}: {e}"[:200]}
        print(json.dumps(line), flush=True)
    if world > 1:
        parallel.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
