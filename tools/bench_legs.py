"""Informational legs of bench.py (round 6: moved out of bench.py so that its default run is the timed region + roofline + parity rows +
cpu_baseline; `python bench.py --legs all` runs them all and adds their figures to the line as flat `config` scalars).

  whole_detector_leg    one clip through the WHOLE detector (PyTorch trunk + this library + GPU post-process + tracker), in the config's
                        head mode (fp16x2, the default) with the bf16 storage policy as a second scalar
  rank_detector_leg     the same per rank at N > 1 (after the timed region)
  single_clip_latency   the hot path on ONE clip (no stacking)
  side_leg              the graph-replayed step in another mode / on another configuration (+ its per-kernel table)
  extra_legs            the loop over all of them, writing into the bench line
None of them is part of `value`.
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0
MFMA_PEAK_TFLOPS = 2500.0


def note(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def whole_detector_leg(a, dev):
    """Informational, rank 0 at N=1, outside the timed region and never part of `value`: one synthetic T-frame clip through
    the WHOLE detector of configs/r50_fpn_slotvps_mi355x.py - ResNet-50 + FPN + semantic tower in PyTorch-ROCm (fp32, as
    the reference runs them; random weights), the slot head and decode of this library (eager, one clip, no stacking), the
    GPU post-process and the tracker (detector.VPS_Temporal_Slots.clip_test). Says what the hot path is a part of."""
    from slotvps_amd.config import Config
    from slotvps_amd.registry import build_detector
    cfg = Config.fromfile(os.path.join(ROOT, "configs", "r50_fpn_slotvps_mi355x.py"))
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

    def timed(fn, n=10):
        fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / n * 1e3, out

    det.use_graph = True                    # the slot head of the clip replays as one hipGraph
    head = det.image_model.dynamic_mask_head
    mode = head.mode                        # the config's mode: fp16x2, the one that meets the tolerance (configs/*_mi355x.py)
    with torch.no_grad():
        t_trunk, (feats, _fcn) = timed(lambda: det.trunk(imgs))
        t_head, _ = timed(lambda: [m.dense() for m in det.head_path(feats)[2:]])
        t_all, res = timed(lambda: det.clip_test(imgs, metas))
        head.set_mode("bf16")               # BASELINE's storage policy (1.4e-2 from the reference: opt-in) as a second scalar
        t_bf16, _ = timed(lambda: det.clip_test(imgs, metas))
        head.set_mode(mode)
    return {"value": round(T / t_all * 1e3, 2), "unit": "frames/s", "ms_per_clip": round(t_all, 2), "timed_iterations": 10, "head_mode": mode,
            "trunk_ms": round(t_trunk, 2), "slot_head_and_all_slot_decode_ms": round(t_head, 2),
            "value_head_mode_bf16": round(T / t_bf16 * 1e3, 2), "ms_per_clip_head_mode_bf16": round(t_bf16, 2),
            "segments_per_frame": [int(len(r["panoptic_cls_inds"])) for r in res],
            "what": f"one {H}x{W} T={T} clip, whole detector: PyTorch fp32 trunk (backbone, FPN, semantic tower with fp32 deformable "
                    f"convolutions) + this library in head mode {mode} (the tower's rows as fp16 hi + lo planes, slot head as one hipGraph, "
                    f"decode of the kept slots only, GPU post-process) + tracker, n_gpus=1; informational, never part of `value`"}


def single_clip_latency(a, dev):
    """Latency of the hot path on ONE clip (no stacking): the same graph-replayed step as the timed region with
    clips_per_launch = 1. The headline `value` stacks 32 clips per launch for throughput; this is what one clip waits."""
    from slotvps_amd.clip import SlotClipRunner
    from slotvps_amd import synth
    r1 = SlotClipRunner(dev, a.frames, a.height, a.width, L=a.slots, param_seed=0, cfg=dict(synth.R50_HEAD_CFG, num_classes=a.num_classes),
                        use_graph=True, n_slots=1, clips_per_launch=1, input_form=a.input_form)
    r1.head.set_mode(a.mode)
    r1.load_clip(r1.random_clip(99))
    for _ in range(3):
        r1.run()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(20):
        r1.run()
    torch.cuda.synchronize(dev)
    return round((time.perf_counter() - t0) / 20 * 1e3, 3)


def rank_detector_leg(a, dev, iters=3):
    """Informational, every rank at N > 1, after the timed region: ms per clip of the WHOLE detector on this rank (PyTorch trunk,
    this library, GPU post-process, tracker + its HOST part) - the per-rank host work the hot-path step does not contain. No
    collective inside: a failure on one rank cannot hang the others (-1 is reported for it)."""
    try:
        from slotvps_amd.config import Config
        from slotvps_amd.registry import build_detector
        cfg = Config.fromfile(os.path.join(ROOT, "configs", "r50_fpn_slotvps_mi355x.py"))
        torch.manual_seed(0)
        det = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg).to(dev).eval()
        T, H, W = a.frames, a.height, a.width
        imgs = torch.randn(T, 3, H, W, device=dev)
        table = torch.zeros(a.slots, 20, device=dev)
        table[torch.arange(a.slots), torch.arange(a.slots) % 19] = 12.0
        with torch.no_grad():
            det.image_model.fg_bn.weight.fill_(40.0)
        base = det.head_path
        det.head_path = lambda f: (lambda lg, em, mk: (lg + table, em, mk))(*base(f))
        metas = [dict(iid=10001 + t, ori_shape=(H, W, 3), img_shape=(H, W, 3), filename=f"f{t}.png") for t in range(T)]
        det.use_graph = True
        with torch.no_grad():
            det.clip_test(imgs, metas)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(iters):
                det.clip_test(imgs, metas)
            torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / iters * 1e3
    except Exception as e:                                   # informational: never costs the bench line
        note(f"rank detector leg failed: {type(e).__name__}: {e}")
        return -1.0


def side_leg(a, dev, frames, height, width, slots, num_classes, cpl, steps, mode="bf16", decode_logits=None, input_form=None, with_roofline=False):
    """frames/s of the same graph-replayed step in another mode / on another configuration (informational legs of the default line).
    with_roofline: the per-kernel table of that step as well (HIP events around every launch of the library, eager pass)."""
    from slotvps_amd.clip import SlotClipRunner
    from slotvps_amd import synth
    form = input_form or ("nchw_f32" if mode in ("fp32", "fp16x2") else "tower16")      # (defaults: each mode's headline form)
    if mode == "fp32":
        form = "nchw_f32"                                                                # (the exact mode takes the reference's fp32 tensors only)
    r1 = SlotClipRunner(dev, frames, height, width, L=slots, param_seed=0, cfg=dict(synth.R50_HEAD_CFG, num_classes=num_classes),
                        use_graph=True, n_slots=1, clips_per_launch=cpl,
                        decode_logits=bool(a.decode_logits) if decode_logits is None else decode_logits, input_form=form)
    r1.head.set_mode(mode)
    r1.load_clip(r1.random_clip(7))
    for _ in range(2):
        r1.run()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        r1.run()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    res = {"value": round(frames * cpl / dt, 2), "unit": "frames/s", "ms_per_step": round(dt * 1e3, 3), "steps": steps,
           "clips_per_launch": cpl, "mode": mode, "input_form": form}
    if with_roofline:
        from slotvps_amd import _lib, ops
        r1.use_graph = False
        r1.run()
        torch.cuda.synchronize(dev)
        kids = {"level_fuse": _lib.KERNEL_LEVEL_FUSE, "retr_stats": _lib.KERNEL_RETR_STATS, "retr_attn": _lib.KERNEL_RETR_ATTN,
                "retr_finish": _lib.KERNEL_RETR_FINISH, "mask_decode": _lib.KERNEL_MASK_DECODE, "kv_project": _lib.KERNEL_KV_PROJECT,
                "slot_attn": _lib.KERNEL_SLOT_ATTN}
        with ops.KernelTimer() as kt:
            for _ in range(steps):
                r1.run()
            torch.cuda.synchronize(dev)
            timed = {name: kt.collect(kid) for name, kid in kids.items()}
        alg = r1.algorithmic_per_step()
        per = {}
        for name, (ms, n) in timed.items():
            if n == 0:
                continue
            e = {"launches": n, "avg_launch_us": round(ms / n * 1e3, 2), "ms_per_step": round(ms / steps, 3)}
            if name in alg:
                sec = ms * 1e-3
                e["algorithmic_bytes_per_launch"] = int(alg[name]["bytes"] * steps / n)
                e["hbm_gbs"] = round(alg[name]["bytes"] * steps / sec / 1e9, 1)
                e["hbm_frac"] = round(e["hbm_gbs"] / HBM_PEAK_GBS, 4)
                e["mfma_tflops"] = round(alg[name]["flops"] * steps / sec / 1e12, 1)
                e["mfma_frac"] = round(e["mfma_tflops"] / MFMA_PEAK_TFLOPS, 4)
                if "executed_flops" in alg[name]:
                    e["mfma_frac_executed"] = round(alg[name]["executed_flops"] * steps / sec / 1e12 / MFMA_PEAK_TFLOPS, 4)
                e["bound"] = "hbm" if e["hbm_frac"] >= e["mfma_frac"] else "mfma"
            per[name] = e
        dom = max((k for k in per if k in alg), key=lambda k: timed[k][0])
        d = per[dom]
        hbm = d["bound"] == "hbm"
        res["roofline"] = {"bound": d["bound"], "achieved": d["hbm_gbs"] if hbm else d["mfma_tflops"],
                           "peak": HBM_PEAK_GBS if hbm else MFMA_PEAK_TFLOPS, "unit": "GB/s" if hbm else "TFLOP/s",
                           "frac": d["hbm_frac"] if hbm else d["mfma_frac"], "traffic": None, "kernel": dom,
                           "what": f"dominant kernel of the {mode} step by device time (HIP events on the launch stream, eager pass)",
                           "avg_launch_us": d["avg_launch_us"], "per_kernel": per,
                           "slot_side_and_rest_ms_per_step": round(dt * 1e3 - sum(v["ms_per_step"] for v in per.values()), 3)}
    return res




def extra_legs(a, dev, line):
    """The informational legs of `python bench.py --legs all` (N = 1, outside the timed region): every figure also as a flat scalar of
    line["config"] (the driver's record keeps the scalars of `config`)."""
    cfgd = line["config"]
    if a.latency_leg:
        note("single-clip latency leg ...")
        try:
            line["single_clip_latency_ms"] = single_clip_latency(a, dev)
        except Exception as e:
            line["single_clip_latency_ms"] = None
            note(f"single-clip latency leg failed: {type(e).__name__}: {e}")
    if a.exact_leg:
        # the other modes of the head on the same step (hipGraph, 3 timed steps each): frames/s + their per-kernel tables
        line["modes"] = {}
        for mode in ("fp16x2", "bf16", "fp16", "fp32"):
            if mode == a.mode:
                continue
            note("exact-mode leg (fp32 storage and arithmetic, one clip per launch) ..." if mode == "fp32" else f"mode leg {mode} ...")
            try:
                ml = side_leg(a, dev, a.frames, a.height, a.width, a.slots, a.num_classes, 1 if mode == "fp32" else a.clips_per_launch, 3,
                              mode=mode, with_roofline=mode != "fp32")
                line["modes"][mode] = ml
                cfgd[f"mode_{mode}_fps"] = ml["value"]
                for kname, e in (ml.get("roofline") or {}).get("per_kernel", {}).items():
                    cfgd[f"mode_{mode}_k_{kname}_ms_per_step"] = e["ms_per_step"]
                    if "hbm_frac" in e:
                        cfgd[f"mode_{mode}_k_{kname}_hbm_frac"] = e["hbm_frac"]
                        cfgd[f"mode_{mode}_k_{kname}_mfma_frac_executed"] = e.get("mfma_frac_executed")
            except Exception as e:
                line["modes"][mode] = {"value": None, "error": f"{type(e).__name__}: {e}"[:200]}
        # rounds 1 - 4's definitions of `value`, for comparison across rounds
        if "bf16" in line["modes"] and line["modes"]["bf16"].get("value") is not None:
            cfgd["value_prev_definition"] = line["modes"]["bf16"]["value"]
            cfgd["value_prev_definition_what"] = "round 4's headline: mode bf16 (BASELINE's storage policy), input_form tower16, fp32 mask logits written"
        for key, kw, what in (("fp16x2_from_the_tower_rows", dict(mode="fp16x2", input_form="tower16"),
                               "round 6: mode fp16x2 from the semantic tower's own rows as two fp16 planes hi + lo, conv_trans composed into K4-HL's weights "
                               "(what the detector runs; `value` starts from the reference head's own fp32 NCHW tensors)"),
                              ("bf16_reference_input_tensors", dict(mode="bf16", input_form="nchw_f32"),
                               "rounds 1 - 3's input: mode bf16 from the reference's fp32 NCHW tensors behind conv_trans"),
                              ("bf16_argmax_only", dict(mode="bf16", decode_logits=False),
                               "round 3's headline workload: mode bf16, K2 in argmax-only mode (the [T, L, HW] fp32 logits are not written)")):
            try:
                fl = side_leg(a, dev, a.frames, a.height, a.width, a.slots, a.num_classes, a.clips_per_launch, 3, **kw)
                fl["what"] = what
                line[key] = fl
                cfgd[f"leg_{key}_fps"] = fl["value"]
            except Exception as e:
                line[key] = {"value": None, "error": f"{type(e).__name__}: {e}"[:200]}
    if a.viper_leg:
        note("VIPER leg (1088x1920 T=10, 200 slots, 24 classes; informational) ...")
        line["other_configs"] = {}
        for mode in ("fp16x2", "bf16"):
            key = f"viper_1088x1920_T10_L200_{mode}"
            try:
                vp = side_leg(a, dev, 10, 1088, 1920, 200, 24, 8, 3, mode=mode)
                vp["what"] = f"BASELINE config 5 geometry on one GPU: 1088x1920 (1080 padded) T=10 clips, 200 slots, 24 classes, 8 clips stacked per launch, mode {mode}, hipGraph"
                line["other_configs"][key] = vp
                cfgd[f"leg_{key}_fps"] = vp["value"]
            except Exception as e:
                line["other_configs"][key] = {"value": None, "error": f"{type(e).__name__}: {e}"[:200]}
    if a.whole_detector:
        note("whole_detector leg (informational) ...")
        try:
            line["whole_detector"] = whole_detector_leg(a, dev)
            cfgd["leg_whole_detector_fps"] = line["whole_detector"].get("value")
            cfgd["leg_whole_detector_head_mode"] = line["whole_detector"].get("head_mode")
            cfgd["leg_whole_detector_head_mode_bf16_fps"] = line["whole_detector"].get("value_head_mode_bf16")
        except Exception as e:                       # informational leg: never costs the bench line
            line["whole_detector"] = {"value": None, "error": f"{type(e).__name__}: {e}"[:200]}
