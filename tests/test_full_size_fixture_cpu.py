"""The full-size fixture (tests/golden/head_full.npz, outputs of the reference's own modules at BASELINE.json's sizes) on the CPU side:
its shape and its recorded reproducibility floor, and the oracle pinned against it where the oracle finishes in seconds - the two coarse
levels (stages 0 - 2: no temporal step, so one frame is independent of the others) of every case."""
import os
import sys

import numpy as np
import pytest

from util import orc, ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import fullsize_parity as fsp  # noqa: E402


def test_fixture_records_the_reference_floor():
    """`floor` = the reference's fp32 run against the SAME modules in float64: what the GPU bounds of tests/test_full_size_gpu.py lean on."""
    z = fsp._Fixtures([fsp.FIXTURE, fsp.FIXTURE_R06])
    assert float(z["T5_1024x2048_L100_floor_mask"]) <= 2e-5 and float(z["T2_1088x1920_L200_floor_mask"]) <= 2e-5
    assert float(z["T10_1088x1920_L200_floor_mask"]) <= 3e-5                  # config 5 at its own clip length (T L = 2000 temporal rows)
    assert 1e-4 <= float(z["T2_1024x2048_L100_sharp_floor_mask"]) <= 2e-3
    for tag in fsp.CASES:
        T, H, W, L, nc = (int(x) for x in z[f"{tag}_meta"][:5])
        assert z[f"{tag}_embeds"].shape == (T, 7, L, 256) and z[f"{tag}_logits"].shape == (T, 7, L, nc)
        assert z[f"{tag}_argmax"].shape == (T, (H // 4) * (W // 4)) and z[f"{tag}_argmax"].max() < L
        assert z[f"{tag}_margin"].dtype == np.float16 and (z[f"{tag}_margin"] >= 0).all()
    # the integer target: panoptic id maps of the reference's own post-process at 1024 x 2048 (frames 0 and T - 1 of the T5 case)
    for t in (0, 4):
        ids = z[f"T5_1024x2048_L100_pan_ids_{t}"]
        assert ids.shape == (1024, 2048) and ids.dtype == np.uint8 and len(np.unique(ids)) >= 10
        assert len(z[f"T5_1024x2048_L100_pan_labels_{t}"]) == len(z[f"T5_1024x2048_L100_pan_slot_index_{t}"]) >= 10
        # round 6: the reference's OWN disagreement on the integer targets, as counts (its fp32 run vs the same modules in float64)
        assert 0 <= int(z[f"T5_1024x2048_L100_floor_pan_diff_pixels_{t}"]) <= 200 and bool(z[f"T5_1024x2048_L100_floor_pan_same_segments_{t}"])
    assert int(z["T5_1024x2048_L100_floor_argmax_diff_pixels"]) == round((1 - float(z["T5_1024x2048_L100_floor_argmax_same"])) * 5 * 131072)
    for tag, n in (("T5_1024x2048_L100", 100), ("T10_1088x1920_L200", 200)):
        f0, f1, dy, dx = (int(x) for x in z[f"{tag}_dense_meta"])
        T, H, W = (int(x) for x in z[f"{tag}_meta"][:3])
        dense = z[f"{tag}_mask_dense"]
        assert (f0, f1) == (0, T - 1) and dense.shape == (2, n, -(-(H // 4) // dy), -(-(W // 4) // dx))
        # the dense sample contains the coarse one (same reference run): frames 0 and T - 1 at the coarse strides
        sy, sx = (int(x) for x in z[f"{tag}_meta"][6:8])
        assert np.array_equal(dense[:, :, ::sy // dy, ::sx // dx], z[f"{tag}_mask_sample"][[0, T - 1]])


@pytest.mark.parametrize("tag", list(fsp.CASES))
def test_oracle_coarse_levels_against_the_full_size_fixture(tag):
    case = fsp.load_case(tag)
    ref = case["ref"]
    cfg = dict(per_level_stages=(1, 2), activation=case["cfg"]["activation"], temporal_activation=case["cfg"]["temporal_activation"])
    feats = [[case["feats"][i][0] for i in range(2)]]                       # frame 0, levels 0 and 1
    pos = [orc.pos_embed_sine(h, w) for (h, w) in case["sizes"][:2]]
    logits, embeds, fused = orc.head_forward(feats, case["slots"], pos, case["params"], cfg=cfg, dt=np.float64)
    s0 = case["strides"][3]
    h0, w0 = case["sizes"][0]
    f0 = fused[0][0].reshape(h0, w0, 256)[::s0, ::s0]
    assert np.abs(f0 - ref["fused0_sample"]).max() <= 2e-5 * max(1.0, np.abs(ref["fused0_sample"]).max())
    for s in range(3):
        e = float(np.abs(embeds[0][s] - ref["embeds"][0, s]).max())
        c = float(np.abs(logits[0][s] - ref["logits"][0, s]).max())
        floor = float(ref["floor_embeds"][s])
        print(f"[{tag}] stage {s}: oracle (float64) vs the reference's fp32 outputs: embeddings {e:.1e}, class logits {c:.1e} (the reference's own fp32 vs float64 {floor:.1e})")
        assert e <= max(3.0 * floor, 5e-5) and c <= max(3.0 * floor, 5e-5)
