"""Pins the post-process oracle (a9) against outputs of the reference's own PostProcessPanopticInstances
(tests/golden/make_golden_post.py). Integer results (kept slots, labels) bit-exact, float masks to 2e-6."""
import os

import numpy as np
import pytest

import synth
from util import GOLDEN, ROOT
import sys
sys.path.insert(0, ROOT)
from oracle import postprocess_oracle as po


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_postprocess_matches_reference(tag):
    z = np.load(os.path.join(GOLDEN, "postprocess.npz"))
    seed, L, h, w, nk = (int(x) for x in z[f"{tag}_meta"])
    logits, masks = synth.make_post_case(seed, L, h, w, 20, nk)
    r = po.postprocess(logits, masks, (4 * h, 4 * w))
    np.testing.assert_array_equal(r["slot_index"], z[f"{tag}_slot_index"])
    np.testing.assert_array_equal(r["labels"], z[f"{tag}_labels"])
    assert np.abs(r["probs"] - z[f"{tag}_probs"]).max() < 1e-6
    assert np.abs(r["masks"] - z[f"{tag}_masks"]).max() < 2e-6
    scores, classes, keep = po.select_slots(logits)
    assert keep.sum() > len(r["labels"])            # the overlap / small-area filters were exercised


def test_relabel_quirks():
    """simple_test :411-435: instance ids counted downward from stuff_num + n_inst - 1; a stuff segment takes
    semantic_labels[position in unique()], not semantic_labels[id]."""
    H, W = 4, 6
    masks = np.full((4, H, W), -5.0, np.float32)
    labels = np.array([12, 3, 15, 7])               # thing, stuff, thing, stuff -> reorder: [3, 7, 12, 15]
    masks[1, :, 0:2] = 1.0                          # stuff 3  -> id 0 after reorder
    masks[0, :, 2:4] = 1.0                          # thing 12 -> id 2
    masks[2, :, 4:6] = 1.0                          # thing 15 -> id 3      (stuff 7 / id 1 owns no pixel)
    out, cls_inds, sem = po.panoptic_relabel(masks, labels)
    assert sem.tolist() == [3, 7, 12, 15] and cls_inds.tolist() == [2, 5]
    assert set(np.unique(out)) == {3, 11, 12}
    assert (out[:, 4:6] == 12).all() and (out[:, 2:4] == 11).all() and (out[:, 0:2] == 3).all()
    # quirk: drop the pixels of stuff id 0 -> unique() = [1?..]: make stuff 7 own pixels instead of stuff 3
    masks2 = masks.copy(); masks2[1] = -5.0; masks2[3, :, 0:2] = 1.0
    out2, _, _ = po.panoptic_relabel(masks2, labels)
    # unique ids = [1, 2, 3]; the stuff region (id 1) sits at position 0 of unique() -> semantic_labels[0] = 3, not 7
    assert (out2[:, 0:2] == 3).all()
