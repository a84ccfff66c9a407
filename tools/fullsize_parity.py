"""Parity of the HIP hot path against the REFERENCE's own outputs at BASELINE.json's sizes (tests/golden/head_full.npz, written by
tests/golden/make_golden_full.py from the imported reference modules). Shared by tests/test_full_size_gpu.py and bench.py - it reads the
fixture (data) and the package only; neither the oracle nor the reference.

    case = load_case("T5_1024x2048_L100")
    row = run_mode(torch.device("cuda:0"), case, "fp16x2")      # dict of scalars, see run_mode

python tools/fullsize_parity.py [--modes ...] [--cases ...] [--out profiles/r05/fullsize_parity.json]   prints / stores the table.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from slotvps_amd import synth  # noqa: E402

FIXTURE = os.path.join(ROOT, "tests", "golden", "head_full.npz")
# round 6 (make_golden_full.py --part r06): BASELINE config 5 at its own clip length (T = 10, 200 slots: the temporal step over 2000 slot
# rows), denser mask-logit samples of frames 0 and T - 1, and the reference's own fp32-vs-float64 disagreement on the integer targets as counts
FIXTURE_R06 = os.path.join(ROOT, "tests", "golden", "head_full_r06.npz")
CASES = ("T5_1024x2048_L100", "T2_1024x2048_L100_sharp", "T2_1088x1920_L200", "T2_1024x2048_L100_swin", "T10_1088x1920_L200")
# the north star's contract: 1e-4 on the float mask logits, the integer slot argmax identical wherever the reference's own top-2 margin
# exceeds DECIDABLE_FACTOR x the measured mask-logit error (below that a pixel's argmax is not determined by values known to +-error)
TOL_MASK = 1e-4
DECIDABLE_FACTOR = 2.0


class _Fixtures:
    """The keys of head_full.npz and head_full_r06.npz as one mapping (a case's keys may be spread over both)."""

    def __init__(self, paths):
        self.zs = [np.load(p) for p in paths if os.path.exists(p)]
        self.files = [k for z in self.zs for k in z.files]

    def __getitem__(self, k):
        for z in self.zs:
            if k in z.files:
                return z[k]
        raise KeyError(k)


def load_case(tag, fixture=None):
    z = _Fixtures([FIXTURE, FIXTURE_R06] if fixture is None else [fixture])
    T, H, W, L, nc, seed, sy, sx, s3, s0 = (int(x) for x in z[f"{tag}_meta"])
    tau = float(z[f"{tag}_tau"])
    import ast
    over = dict(ast.literal_eval(str(z[f"{tag}_cfg"]))) if f"{tag}_cfg" in z.files else {}      # head-config overrides of the case (Swin-L head)
    cfg = dict(synth.R50_HEAD_CFG, num_classes=nc, **over)
    params = synth.temper_queries(synth.make_params(synth.head_shapes(cfg), seed), tau)
    feats = synth.make_clip_features(seed + 1, T, H, W)
    feats = [np.stack([feats[t][i] for t in range(T)]) for i in range(4)]          # per level [T, 128, h, w]
    bn, fg = synth.make_feat_bn(seed + 3)
    return dict(tag=tag, T=T, H=H, W=W, L=L, nc=nc, seed=seed, tau=tau, cfg=cfg, params=params, feats=feats,
                slots=synth.make_slots(seed + 2, L), sizes=synth.level_sizes(H, W), bn=bn, fg=fg, strides=(sy, sx, s3, s0),
                ref={k[len(tag) + 1:]: z[k] for k in z.files if k.startswith(tag + "_")})


def build_head(dev, case, mode):
    import torch
    from slotvps_amd.clip import build_r50_head
    head = build_r50_head(case["cfg"])
    sd = head.state_dict()
    head.load_state_dict({k: torch.from_numpy(v).reshape(sd[k].shape) for k, v in case["params"].items()}, strict=True)
    return head.to(dev).eval().set_mode(mode)


def _bns(dev, case):
    import torch
    feat_bn = torch.nn.BatchNorm2d(256).to(dev).eval()
    fg_bn = torch.nn.BatchNorm2d(1).to(dev).eval()
    w, b, mu, var = case["bn"]
    fg = case["fg"]
    with torch.no_grad():
        feat_bn.weight.copy_(torch.from_numpy(w)); feat_bn.bias.copy_(torch.from_numpy(b))
        feat_bn.running_mean.copy_(torch.from_numpy(mu)); feat_bn.running_var.copy_(torch.from_numpy(var))
        fg_bn.weight.fill_(float(fg[0])); fg_bn.bias.fill_(float(fg[1]))
        fg_bn.running_mean.fill_(float(fg[2])); fg_bn.running_var.fill_(float(fg[3]))
    return feat_bn, fg_bn


def _map_f32(f):
    """fused level map of any mode -> [T, HW, 256] fp32 (fp16x2: the sum of the hi and lo planes)."""
    return f[0].float() + f[1].float() if f.dim() == 4 else f.float()


def run_mode(dev, case, mode, teacher_forced=True):
    """One clip of `case` through head.forward_clip + generate_final_outputs in `mode`, against the reference's outputs.
    Returns scalars: free_embed_err[7], free_logit_err[7], tf_embed_err[7] (stage s fed the REFERENCE's stage s-1 embeddings),
    mask_err (sampled pixels, every frame, free-running; mask_err_dense: the 16 x denser sample of frames 0 and T - 1 where the
    fixture holds one - `meets` asks both), argmax_diff_pixels / pixels (COUNTS, beside ref_floor_argmax_diff_pixels: the reference's own
    fp32 run against the same modules in float64), mask_err_tf (decode of the reference's own last-stage embeddings on this
    mode's map), argmax_equal (fraction of all pixels, free-running), decidable (fraction of pixels whose reference margin exceeds
    DECIDABLE_FACTOR x mask_err), argmax_equal_decidable, fused3_err / fused0_err (relative to the map's largest magnitude), meets."""
    import torch
    from slotvps_amd import ops
    from slotvps_amd.slot_head import generate_final_outputs
    T, L, sizes, ref = case["T"], case["L"], case["sizes"], case["ref"]
    sy, sx, s3, s0 = case["strides"]
    head = build_head(dev, case, mode)
    feat_bn, fg_bn = _bns(dev, case)
    cfgh = case["cfg"]
    with torch.no_grad():
        tf = [torch.from_numpy(f).to(dev) for f in case["feats"]]
        pos_tabs = [ops.pos_embed_sine_tables(h, w, 256, dev) for (h, w) in sizes]
        slots = torch.from_numpy(case["slots"]).to(dev)
        logits, embeds, fused = head.forward_clip(tf, slots, pos_tabs)
        masks, amax = generate_final_outputs(fused[3], embeds[6].contiguous(), feat_bn, fg_bn, want_argmax=True)
        emb_ref = torch.from_numpy(np.ascontiguousarray(ref["embeds"][:, 6])).to(dev)
        masks_tf = generate_final_outputs(fused[3], emb_ref, feat_bn, fg_bn)
        torch.cuda.synchronize()
        h3, w3 = sizes[3]
        E = embeds.cpu().numpy()                                   # [7, T, L, 256]
        C = logits.cpu().numpy()
        samp = masks.view(T, L, h3, w3)[:, :, ::sy, ::sx].cpu().numpy()
        samp_tf = masks_tf.view(T, L, h3, w3)[:, :, ::sy, ::sx].cpu().numpy()
        am = amax.cpu().numpy().reshape(T, -1)
        # the uint8 argmax the kernel wrote must be the argmax of the logits it wrote (ties: lowest slot, like torch.argmax on the host side of the reference)
        am_of_logits = masks.argmax(dim=1).cpu().numpy().reshape(T, -1)
        f3 = _map_f32(fused[3])[T - 1].view(h3, w3, 256)[::s3, ::s3].cpu().numpy()
        h0, w0 = sizes[0]
        f0 = _map_f32(fused[0])[0].view(h0, w0, 256)[::s0, ::s0].cpu().numpy()
        row = dict(mode=mode, case=case["tag"])
        row["free_embed_err"] = [float(np.abs(E[s].astype(np.float64) - ref["embeds"][:, s]).max()) for s in range(7)]
        row["free_logit_err"] = [float(np.abs(C[s].astype(np.float64) - ref["logits"][:, s]).max()) for s in range(7)]
        row["mask_err"] = float(np.abs(samp.astype(np.float64) - ref["mask_sample"]).max())
        row["mask_err_tf"] = float(np.abs(samp_tf.astype(np.float64) - ref["mask_sample"]).max())
        row["mask_err_dense"] = None
        if "mask_dense" in ref:                                    # frames (f0, f1) at stride (dy, dx): 16 x the pixels of mask_sample
            f0_, f1_, dy, dx = (int(x) for x in ref["dense_meta"])
            dense = masks.view(T, L, h3, w3)[[f0_, f1_]][:, :, ::dy, ::dx].cpu().numpy()
            row["mask_err_dense"] = float(np.abs(dense.astype(np.float64) - ref["mask_dense"]).max())
            row["mask_dense_samples"] = int(dense.size)
        row["mask_err_max"] = max(row["mask_err"], row["mask_err_dense"] or 0.0)
        same = am == ref["argmax"]
        margin = ref["margin"].astype(np.float64)
        dec = margin > DECIDABLE_FACTOR * row["mask_err_max"]
        # the integer target as COUNTS, beside the reference's own disagreement with itself (its fp32 run vs the same modules in float64)
        row["pixels"] = int(same.size)
        row["argmax_diff_pixels"] = int((~same).sum())
        row["ref_floor_argmax_diff_pixels"] = (int(ref["floor_argmax_diff_pixels"]) if "floor_argmax_diff_pixels" in ref
                                               else int(round((1.0 - float(ref["floor_argmax_same"])) * same.size)))
        row["argmax_equal"] = float(same.mean())
        row["decidable"] = float(dec.mean())
        row["argmax_equal_decidable"] = float(same[dec].mean()) if dec.any() else 1.0
        row["argmax_kernel_vs_own_logits"] = float((am == am_of_logits).mean())
        row["fused3_err"] = float(np.abs(f3 - ref["fused3_sample"]).max() / float(ref["fused3_absmax"]))
        row["fused0_err"] = float(np.abs(f0 - ref["fused0_sample"]).max() / max(1.0, float(np.abs(ref["fused0_sample"]).max())))
        if teacher_forced:
            tf_err = []
            sidx = 0
            for lvl, n in enumerate(cfgh["per_dh_num_heads"]):
                h, w = sizes[lvl]
                for j in range(n):
                    s_in = np.broadcast_to(case["slots"], (T, L, 256)) if sidx == 0 else ref["embeds"][:, sidx - 1]
                    stage = getattr(head, f"head_series_{lvl}")[j]
                    _, em = stage.forward_pm(torch.from_numpy(np.ascontiguousarray(s_in, dtype=np.float32)).to(dev), fused[lvl], (h, w), pos_tabs[lvl],
                                             sidx in cfgh["apply_temporal_query_atten_stages"], 1)
                    tf_err.append(float(np.abs(em.cpu().numpy().astype(np.float64) - ref["embeds"][:, sidx]).max()))
                    sidx += 1
            row["tf_embed_err"] = tf_err
    row["ref_floor_mask"] = float(ref["floor_mask"])
    row["ref_floor_embeds"] = [float(x) for x in ref["floor_embeds"]]
    row["meets"] = bool(row["mask_err_max"] <= TOL_MASK and row["argmax_equal_decidable"] == 1.0)
    del head, fused, masks, masks_tf, tf
    torch.cuda.empty_cache()
    return row


POST_CFG = dict(is_thing_map={i: i > 10 for i in range(20)}, threshold=0.85, fraction_threshold=0.03, pixel_threshold=0.4,
                apply_mask_removal=True, apply_mask_removal_only_ins=True, use_mask_low_constant=False)


def panoptic_rows(dev, case, mode):
    """The INTEGER target at full size: free-running head in `mode` -> decode -> panoptic post-process (K6) -> relabel, against the id maps
    the REFERENCE's own PostProcessPanopticInstances + the relabel of simple_test produced from the reference head's outputs at
    1024 x 2048 (fixture keys pan_ids_<t>: frames 0 and T - 1; class bias / mask gain: synth.full_size_class_bias, FULL_SIZE_MASK_GAIN).
    Returns per frame: fraction of pixels with the reference's panoptic id, kept slots equal, labels equal."""
    import torch
    from slotvps_amd import ops
    from slotvps_amd.slot_head import generate_final_outputs
    from slotvps_amd.postprocess import PostProcessPanopticInstances
    T, L, H, W, sizes, ref = case["T"], case["L"], case["H"], case["W"], case["sizes"], case["ref"]
    h3, w3 = sizes[3]
    head = build_head(dev, case, mode)
    feat_bn, fg_bn = _bns(dev, case)
    bias = torch.from_numpy(synth.full_size_class_bias(L, case["nc"])).to(dev)
    rows = []
    with torch.no_grad():
        tf = [torch.from_numpy(f).to(dev) for f in case["feats"]]
        pos_tabs = [ops.pos_embed_sine_tables(h, w, 256, dev) for (h, w) in sizes]
        logits, embeds, fused = head.forward_clip(tf, torch.from_numpy(case["slots"]).to(dev), pos_tabs)
        masks = generate_final_outputs(fused[3], embeds[6].contiguous(), feat_bn, fg_bn)
        pp = PostProcessPanopticInstances(**POST_CFG)
        for t in sorted(int(k.split("_")[-1]) for k in ref if k.startswith("pan_ids_")):
            res = pp.forward_tensors(logits[6, t] + bias, (synth.FULL_SIZE_MASK_GAIN * masks[t]).view(L, h3, w3).contiguous(), (H, W))
            ids, cls_inds, _ = pp.panoptic_ids(res)
            torch.cuda.synchronize()
            ids = ids.cpu().numpy().reshape(H, W).astype(np.int64)
            want = ref[f"pan_ids_{t}"].astype(np.int64)
            rows.append(dict(mode=mode, frame=t, ids_equal=float((ids == want).mean()), pixels=int(ids.size), ids_diff_pixels=int((ids != want).sum()),
                             # the reference's own post-process on its float64 outputs against the same on its fp32 outputs (round 6)
                             ref_floor_ids_diff_pixels=int(ref[f"floor_pan_diff_pixels_{t}"]) if f"floor_pan_diff_pixels_{t}" in ref else None,
                             slots_equal=bool(np.array_equal(res.slot_index.cpu().numpy(), ref[f"pan_slot_index_{t}"])),
                             labels_equal=bool(np.array_equal(res.labels.cpu().numpy(), ref[f"pan_labels_{t}"])),
                             segments=int(len(ref[f"pan_labels_{t}"]))))
    del head, fused, masks, tf
    torch.cuda.empty_cache()
    return rows


def fmt(row):
    e = lambda xs: " ".join(f"{x:.1e}" for x in xs)
    dense = "" if row.get("mask_err_dense") is None else f", {row['mask_err_dense']:.2e} on the dense sample ({row['mask_dense_samples']} logits)"
    s = (f"[{row['case']} / {row['mode']}] mask logits {row['mask_err']:.2e} free-running{dense} ({row['mask_err_tf']:.2e} with the reference's last embeddings; "
         f"the reference's own fp32 vs float64: {row['ref_floor_mask']:.1e}); slot argmax differs on {row['argmax_diff_pixels']} of {row['pixels']} pixels "
         f"(the reference's fp32 run vs its float64 run: {row['ref_floor_argmax_diff_pixels']}) = equal on {100 * row['argmax_equal']:.4f} %, "
         f"{100 * row['argmax_equal_decidable']:.4f} % of the {100 * row['decidable']:.2f} % decidable; fused maps {row['fused0_err']:.1e} / {row['fused3_err']:.1e}\n"
         f"    embeddings per stage, free-running  {e(row['free_embed_err'])}\n")
    if "tf_embed_err" in row:
        s += f"    embeddings per stage, teacher-forced {e(row['tf_embed_err'])}\n"
    s += f"    the reference's fp32 vs float64      {e(row['ref_floor_embeds'])}    meets 1e-4 / argmax: {row['meets']}"
    return s


def main():
    import torch
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", default="bf16,fp16,fp16x2,fp32")
    ap.add_argument("--cases", default=",".join(CASES))
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    rows = []
    for tag in a.cases.split(","):
        case = load_case(tag)
        for mode in a.modes.split(","):
            try:
                row = run_mode(dev, case, mode)
                print(fmt(row), flush=True)
            except NotImplementedError as e:
                row = dict(mode=mode, case=tag, error=str(e))
                print(f"[{tag} / {mode}] not implemented: {e}", flush=True)
            rows.append(row)
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
