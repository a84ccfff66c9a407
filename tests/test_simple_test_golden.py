"""Test-time flow after the head - post-process, relabel (vps_temporal_slots.py:411-435), tracker assignment (:328-409),
result dict - against the reference's OWN simple_test run on a four-frame synthetic video (tests/golden/simple_test.npz,
made by tests/golden/make_golden_flow.py). CPU: pins the oracle restatements (postprocess_oracle.panoptic_relabel /
track_assign were unpinned before). GPU: the product (K6 + host tables + tracker) on the same head outputs."""
import os

import numpy as np
import pytest
import torch

from oracle import postprocess_oracle as porc
from slotvps_amd import synth
from util import GOLDEN


def _case():
    z = np.load(os.path.join(GOLDEN, "simple_test.npz"))
    seed, n_frames, L, lh, lw = (int(v) for v in z["meta"])
    frames, fc_w, fc_b = synth.make_simple_test_case(seed, n_frames, L, lh, lw)
    return z, frames, fc_w, fc_b, (4 * lh, 4 * lw)


def test_oracle_pipeline_equals_reference_simple_test():
    z, frames, fc_w, fc_b, size = _case()
    memory, matched, fresh = None, 0, 0
    for f, fr in enumerate(frames):
        want = porc.postprocess(fr["logits"], fr["masks"], size)
        pan, cls_inds, _ = porc.panoptic_relabel(want["masks"], want["labels"])
        emb = fr["embed"][want["slot_index"]]
        if memory is None:
            det_ids, memory = np.arange(len(emb)), emb.copy()
        else:
            n_before = len(memory)
            det_ids, memory = porc.track_assign(emb, memory, fc_w, fc_b)
            matched += int((det_ids < n_before).sum())
            fresh += int((det_ids >= n_before).sum())
        ins = want["labels"] > 10
        np.testing.assert_array_equal(pan, z[f"f{f}_panoptic_outputs"][0])
        assert cls_inds.tolist() == z[f"f{f}_panoptic_cls_inds"].tolist()
        assert det_ids[ins].tolist() == z[f"f{f}_panoptic_det_obj_ids"].tolist()
        np.testing.assert_allclose(want["probs"][ins], z[f"f{f}_panoptic_cls_prob"], rtol=1e-6)
        np.testing.assert_array_equal(fr["fcn"].argmax(0), z[f"f{f}_fcn_outputs"][0])
    np.testing.assert_array_equal(memory, z["memory"])
    assert matched > 0 and fresh > 0, "the fixture must exercise both re-identification and new identities"


@pytest.mark.gpu
def test_detector_flow_equals_reference_simple_test():
    from test_detector import build
    z, frames, fc_w, fc_b, (H, W) = _case()
    dev = torch.device("cuda:0")
    det = build().to(dev).eval()
    with torch.no_grad():
        for fc, w_, b_ in zip(det.temporal_track_head.fcs_query, fc_w, fc_b):
            fc.weight.copy_(torch.from_numpy(w_))
            fc.bias.copy_(torch.from_numpy(b_))
    stack = lambda k: torch.from_numpy(np.stack([fr[k] for fr in frames])).to(dev)
    frozen = (stack("logits"), stack("embed"), stack("masks"), stack("fcn"))
    det.slot_path = lambda _imgs: frozen
    T = len(frames)
    imgs = torch.zeros(T, 3, H, W, device=dev)
    metas = [dict(iid=30001 + t, ori_shape=(H, W, 3), img_shape=(H, W, 3), filename=f"v3_f{t + 1}.png") for t in range(T)]
    results = det.clip_test(imgs, metas)
    for f, got in enumerate(results):
        np.testing.assert_array_equal(got["panoptic_outputs"].cpu().numpy().astype(np.uint8), z[f"f{f}_panoptic_outputs"])
        np.testing.assert_array_equal(got["fcn_outputs"].cpu().numpy().astype(np.uint8), z[f"f{f}_fcn_outputs"])
        assert got["panoptic_cls_inds"].tolist() == z[f"f{f}_panoptic_cls_inds"].tolist()
        assert got["panoptic_det_obj_ids"].tolist() == z[f"f{f}_panoptic_det_obj_ids"].tolist()
        np.testing.assert_allclose(got["panoptic_cls_prob"].cpu().numpy(), z[f"f{f}_panoptic_cls_prob"], rtol=1e-6)
    np.testing.assert_array_equal(det.prev_embedding.cpu().numpy(), z["memory"])


def test_oracle_mask_decode_equals_reference_generate_final_outputs():
    """a8 against the reference's own method (vps_temporal_slots.py:144-160), not a re-execution of its torch ops."""
    from util import orc
    z = np.load(os.path.join(GOLDEN, "simple_test.npz"))
    case = synth.make_decode_case(int(z["decode_seed"][0]))
    D, h, w = case["feat"].shape
    scale, shift = orc.bn_eval_affine(*case["feat_bn"])
    fgs, fgb = orc.bn_eval_affine(*case["fg_bn"])
    m = orc.mask_decode(case["feat"].reshape(D, h * w).T, case["embed"], scale, shift, float(fgs[0]), float(fgb[0]))
    ref = z["decode_mask"].reshape(len(case["embed"]), h * w)
    assert m.shape == ref.shape and np.abs(m - ref).max() < 2e-6
