"""Parity at BASELINE.json's sizes against the REFERENCE ITSELF (tests/golden/head_full.npz: outputs of the imported
MultiScaleDynamicMaskHead, dynamic_mask_head.py:138-228, and of generate_final_outputs, vps_temporal_slots.py:144-160, at
1024 x 2048 T = 5 / 100 slots, at the VIPER geometry 1088 x 1920 / 200 slots and with the Swin-L config's head; tests/golden/make_golden_full.py).

Every mode of the head (MultiScaleDynamicMaskHead.MODES) runs the whole hot path - four level fusions, seven stages, decode of every
frame - free-running and teacher-forced per stage (stage s fed the reference's stage s - 1 embeddings) through the C ABI; the numbers
are printed and held to the bounds below (tools/fullsize_parity.py computes them; bench.py reports the same rows).

The contract (north star): mask logits within 1e-4 of the reference's, slot argmax identical wherever decidable (reference margin >
2 x the measured error). It is asserted FREE-RUNNING for the modes that claim it (fp16x2, fp32) on the tempered cases, whose chain is
reproducible (the reference's own fp32 result sits 5e-6 from its float64 evaluation). On the `sharp` case (the untempered synth weights:
logit sigma ~ 16, every stage amplifies 2 - 4 x) the reference's own fp32 result is 4.3e-4 from its float64 evaluation - there the
bound is that floor (x 3), and the per-stage teacher-forced errors carry the claim. The 16-bit modes (bf16: BASELINE's storage policy;
fp16: same bytes, three more mantissa bits) cannot meet 1e-4 - the rounding of the stored level maps alone moves the mask logits by
1.1e-3 / 1.5e-4 - and are held to their measured behaviour as regression bounds."""
import os
import sys

import numpy as np
import pytest

from util import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import fullsize_parity as fsp  # noqa: E402

pytestmark = pytest.mark.gpu

_cases = {}


def _case(tag):
    if tag not in _cases:
        _cases.clear()                                # one case's inputs (~0.5 GB of host arrays) at a time
        _cases[tag] = fsp.load_case(tag)
    return _cases[tag]


# (mode, case) -> bounds: mask = free-running mask logits (sampled pixels, all frames); tf = largest teacher-forced per-stage embedding
# error; eq = fraction of ALL pixels with the reference's slot argmax (free-running); contract = must meet 1e-4 / argmax-where-decidable.
# Measured (round 6, MI355X):  fp16x2  T5 9.5e-6 (dense sample 9.5e-6) / 28 of 655 360 argmax pixels (the reference's own fp32 vs float64: 10)
#                                      VIPER T10 6.5e-6 / 45 of 1 305 600 (reference: 27); sharp 4.1e-4 (floor 4.3e-4) / 337 (reference: 197)
# Measured (round 5, MI355X):  fp16x2  T5 9.7e-6 / 3.4e-5 / 99.996 %   sharp 4.0e-4 / 3.9e-5 / 99.86 %   VIPER see test
#                              fp32    T5 5.0e-6 / 1.2e-5 / 99.999 %   sharp 3.9e-4 / 2.6e-5 / 99.91 %   VIPER 3.7e-6 / - / 99.997 %
#                              fp16    T5 1.9e-3 / 3.7e-3 / 99.36 %    sharp 7.8e-2 / 1.3e-2 / 79 %      VIPER 1.6e-3 / 3.0e-3 / 99.32 %
#                              bf16    T5 1.4e-2 / 2.9e-2 / 94.8 %     sharp 1.9e-1 / 1.0e-1 / 35 %      VIPER 1.2e-2 / 2.6e-2 / 94.6 %
BOUNDS = {
    ("fp16x2", "T5_1024x2048_L100"): dict(mask=1e-4, tf=1e-4, eq=0.9995, contract=True),
    ("fp32", "T5_1024x2048_L100"): dict(mask=1e-4, tf=1e-4, eq=0.9995, contract=True),
    ("fp16", "T5_1024x2048_L100"): dict(mask=4e-3, tf=8e-3, eq=0.99, contract=False),
    ("bf16", "T5_1024x2048_L100"): dict(mask=3e-2, tf=6e-2, eq=0.93, contract=False),
    ("fp16x2", "T2_1024x2048_L100_sharp"): dict(mask=None, tf=1e-4, eq=0.995, contract=False, dec=0.9),
    ("fp32", "T2_1024x2048_L100_sharp"): dict(mask=None, tf=1e-4, eq=0.995, contract=False, dec=0.9),
    ("fp16", "T2_1024x2048_L100_sharp"): dict(mask=2e-1, tf=3e-2, eq=0.6, contract=False),
    ("bf16", "T2_1024x2048_L100_sharp"): dict(mask=4e-1, tf=2e-1, eq=0.25, contract=False),
    ("fp16x2", "T2_1088x1920_L200"): dict(mask=1e-4, tf=1e-4, eq=0.9995, contract=True),
    ("fp32", "T2_1088x1920_L200"): dict(mask=1e-4, tf=1e-4, eq=0.9995, contract=True),
    ("fp16", "T2_1088x1920_L200"): dict(mask=4e-3, tf=8e-3, eq=0.99, contract=False),
    ("bf16", "T2_1088x1920_L200"): dict(mask=3e-2, tf=6e-2, eq=0.93, contract=False),
    # the Swin-L config's head (ReLU feed-forward block, GELU temporal head; BASELINE config 4)
    ("fp16x2", "T2_1024x2048_L100_swin"): dict(mask=1e-4, tf=1e-4, eq=0.9995, contract=True),
    ("fp32", "T2_1024x2048_L100_swin"): dict(mask=1e-4, tf=1e-4, eq=0.9995, contract=True),
    ("fp16", "T2_1024x2048_L100_swin"): dict(mask=4e-3, tf=8e-3, eq=0.99, contract=False),
    ("bf16", "T2_1024x2048_L100_swin"): dict(mask=3e-2, tf=6e-2, eq=0.93, contract=False),
    # BASELINE config 5 at its OWN clip length (round 6): T = 10, 200 slots - the temporal step attends over T L = 2000 slot rows
    # (dynamic_mask_head.py:559-567), more than 128 slots in every retriever launch. The reference's own fp32 run sits 1.5e-5 (mask logits) /
    # 2.8e-4 (last-stage embeddings) from its float64 evaluation here
    ("fp16x2", "T10_1088x1920_L200"): dict(mask=1e-4, tf=1e-4, eq=0.9995, contract=True),
    ("fp32", "T10_1088x1920_L200"): dict(mask=1e-4, tf=1e-4, eq=0.9995, contract=True),
    ("fp16", "T10_1088x1920_L200"): dict(mask=4e-3, tf=8e-3, eq=0.99, contract=False),
    ("bf16", "T10_1088x1920_L200"): dict(mask=3e-2, tf=6e-2, eq=0.93, contract=False),
}


@pytest.mark.parametrize("tag", list(fsp.CASES))
@pytest.mark.parametrize("mode", ["fp16x2", "fp32", "fp16", "bf16"])
def test_full_size_clip_against_the_reference(cuda, tag, mode):
    case = _case(tag)
    b = BOUNDS[(mode, tag)]
    row = fsp.run_mode(cuda, case, mode)
    print("\n" + fsp.fmt(row))
    floor = row["ref_floor_mask"]
    # the level maps themselves (K4): fp32-class in the two reference-precision modes, one storage rounding otherwise
    fused_bound = {"fp16x2": 5e-6, "fp32": 5e-6, "fp16": 1e-3, "bf16": 8e-3}[mode]
    assert row["fused0_err"] <= fused_bound and row["fused3_err"] <= fused_bound
    assert max(row["tf_embed_err"]) <= b["tf"], row["tf_embed_err"]
    assert row["free_embed_err"][0] <= b["tf"]                          # stage 0 free-running == teacher-forced
    mask_bound = b["mask"] if b["mask"] is not None else 3.0 * floor     # sharp case: the reference's own fp32-vs-float64 distance
    assert row["mask_err"] <= mask_bound, (row["mask_err"], mask_bound)
    if row["mask_err_dense"] is not None:                                # the 16 x denser sample of frames 0 and T - 1 (round 6) under the same bound
        assert row["mask_err_dense"] <= mask_bound, (row["mask_err_dense"], mask_bound)
    # the integer target as a COUNT: the modes that claim the contract may disagree with the reference on a few times as many pixels as the
    # reference's own fp32 run disagrees with its float64 evaluation (those pixels sit on a top-2 tie at the modes' 1e-5-class error)
    if b["contract"]:
        assert row["argmax_diff_pixels"] <= max(20, 10 * row["ref_floor_argmax_diff_pixels"]), (row["argmax_diff_pixels"], row["ref_floor_argmax_diff_pixels"])
    assert row["argmax_equal"] >= b["eq"], row["argmax_equal"]
    # integer target: identical wherever decidable at the measured error - claimed only where that set is (nearly) everything (ADVICE r05:
    # the threshold derives from the measured error, so for the 16-bit modes the set shrinks to 0 - 60 % of the pixels and the statement would
    # be vacuous; they are held to the `eq` regression bound on ALL pixels instead). Measured decidable: contract rows 99.7 - 99.9 %, sharp 93 %
    dec_floor = 0.99 if b["contract"] else b.get("dec")
    if dec_floor is not None:
        assert row["decidable"] >= dec_floor and row["argmax_equal_decidable"] == 1.0, (row["decidable"], row["argmax_equal_decidable"])
    assert row["argmax_kernel_vs_own_logits"] >= 0.99999                 # the fused argmax byte is the argmax of the logits written
    if b["contract"]:
        assert row["meets"] and row["mask_err"] <= fsp.TOL_MASK
    if mode in ("fp16x2", "fp32"):
        # decode alone (the reference's own last-stage embeddings on this mode's finest map): fp32 summation order
        assert row["mask_err_tf"] <= 2e-6


@pytest.mark.parametrize("mode,bound", [("fp16x2", 0.9995), ("fp32", 0.9995), ("fp16", 0.99), ("bf16", 0.75)])
def test_full_size_panoptic_ids_against_the_reference(cuda, mode, bound):
    """The INTEGER target at BASELINE's size: free-running head -> decode (K2) -> post-process on the device (K6) -> relabel against the
    panoptic id maps the REFERENCE's own PostProcessPanopticInstances (vps_temporal_slots.py:528-807) + the relabel of simple_test
    (:411-435) produced from the reference head's own outputs at 1024 x 2048 (frames 0 and T - 1 of the T5 case; 25 segments: stuff with
    duplicated classes and things). The modes that meet the float tolerance must keep the same slots with the same labels and agree on
    (almost) every pixel - a pixel can differ only where two scaled mask logits lie within the modes' 1e-5-class error of each other;
    the 16-bit modes are held to their measured agreement. Measured (round 5): fp16x2 99.998 / 99.997 %, fp32 99.999 / 99.998 %, fp16
    99.68 / 99.60 % (same segments), bf16 87.5 / 82.8 % (frame 4: a different set of kept segments)."""
    case = _case("T5_1024x2048_L100")
    rows = fsp.panoptic_rows(cuda, case, mode)
    assert len(rows) == 2
    for r in rows:
        print(f"\n[{mode}] frame {r['frame']}: panoptic ids differ on {r['ids_diff_pixels']} of the {r['pixels']} pixels (the reference's own fp32 run vs "
              f"its float64 run: {r['ref_floor_ids_diff_pixels']}) = equal on {100 * r['ids_equal']:.4f} %; kept slots equal "
              f"{r['slots_equal']}, labels equal {r['labels_equal']} ({r['segments']} segments)")
        assert r["ids_equal"] >= bound, r
        if mode in ("fp16x2", "fp32"):                                   # a small multiple of the reference's own disagreement (15 / 35 pixels)
            assert r["ids_diff_pixels"] <= 10 * max(10, r["ref_floor_ids_diff_pixels"]), r
        if mode in ("fp16x2", "fp32"):
            assert r["slots_equal"] and r["labels_equal"], r
