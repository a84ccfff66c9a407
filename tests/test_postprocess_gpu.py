"""K6 parity: the GPU panoptic post-process vs (i) the golden outputs of the reference's own
PostProcessPanopticInstances and (ii) the CPU oracle, on the same inputs. Integer results (kept slots,
labels, per-pixel panoptic ids) must be bit-exact; the materialised float masks within 2e-6."""
import os
import sys

import numpy as np
import pytest

import synth
from util import GOLDEN, ROOT
sys.path.insert(0, ROOT)
from oracle import postprocess_oracle as po

pytestmark = pytest.mark.gpu

CFG = dict(is_thing_map={i: i > 10 for i in range(20)}, threshold=0.85, fraction_threshold=0.03, pixel_threshold=0.4,
           apply_mask_removal=True, apply_mask_removal_only_ins=True, use_mask_low_constant=False)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_postprocess_matches_reference_and_oracle(cuda, tag):
    import torch
    from slotvps_amd.postprocess import PostProcessPanopticInstances
    z = np.load(os.path.join(GOLDEN, "postprocess.npz"))
    seed, L, h, w, nk = (int(x) for x in z[f"{tag}_meta"])
    logits, masks = synth.make_post_case(seed, L, h, w, 20, nk)
    pp = PostProcessPanopticInstances(**CFG)
    res = pp.forward_tensors(torch.from_numpy(logits).to(cuda), torch.from_numpy(masks).to(cuda), (4 * h, 4 * w),
                             materialize_masks=True)
    torch.cuda.synchronize()
    # (i) the reference's outputs
    np.testing.assert_array_equal(res.slot_index.cpu().numpy(), z[f"{tag}_slot_index"])
    np.testing.assert_array_equal(res.labels.cpu().numpy(), z[f"{tag}_labels"])
    assert np.abs(res.probs.cpu().numpy() - z[f"{tag}_probs"]).max() < 1e-6
    got_masks = res.masks.cpu().numpy()
    assert np.abs(got_masks - z[f"{tag}_masks"]).max() < 2e-6
    # (ii) per-pixel panoptic ids vs the oracle relabel of the REFERENCE's masks: bit-exact
    want_ids, want_cls, _ = po.panoptic_relabel(z[f"{tag}_masks"], z[f"{tag}_labels"])
    ids, cls_inds, probs = pp.panoptic_ids(res)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(ids.cpu().numpy().astype(np.int64), want_ids)
    np.testing.assert_array_equal(cls_inds.numpy(), want_cls)
    # upsampling is bit-identical to the oracle's (no fused multiply-add in K6)
    o = po.postprocess(logits, masks, (4 * h, 4 * w))
    stuff = o["labels"] <= 10
    np.testing.assert_array_equal(got_masks[stuff], o["masks"][stuff])


def test_postprocess_full_size_properties(cuda):
    """1024x2048 output from 256x512 logits (BASELINE size): ids only take values of kept segments, areas of
    the final argmax sum to H*W, every surviving segment owns > 4 pixels, result is deterministic."""
    import torch
    from slotvps_amd.postprocess import PostProcessPanopticInstances
    logits, masks = synth.make_post_case(21, 100, 256, 512, 20, 30)
    pp = PostProcessPanopticInstances(**CFG)
    tl, tm = torch.from_numpy(logits).to(cuda), torch.from_numpy(masks).to(cuda)
    res = pp.forward_tensors(tl, tm, (1024, 2048))
    ids, cls_inds, _ = pp.panoptic_ids(res)
    ids2, _, _ = pp.panoptic_ids(pp.forward_tensors(tl, tm, (1024, 2048)))
    torch.cuda.synchronize()
    assert torch.equal(ids, ids2)
    assert ids.shape == (1024, 2048)
    assert all(a > 4 for a in res.area) and sum(res.area) == 1024 * 2048
    labels = res.labels.cpu().numpy()
    n_inst = int((labels > 10).sum())
    allowed = set(int(x) for x in labels[labels <= 10]) | set(range(11, 11 + n_inst))
    assert set(np.unique(ids.cpu().numpy()).tolist()) <= allowed
