"""Harness glue: get_unified_pan_result against outputs of the reference's own method
(tests/golden/harness.npz, made by tests/golden/make_golden_harness.py) and the single_gpu_test result layout."""
import os

import numpy as np
import pytest
import torch

from slotvps_amd import harness, synth
from util import GOLDEN


@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("with_ids", [True, False])
def test_unified_pan_result_matches_reference(seed, with_ids):
    z = np.load(os.path.join(GOLDEN, "harness.npz"))
    segs, pans, cls_inds, obj_ids, names = synth.make_harness_case(seed)
    got = harness.get_unified_pan_result(segs, pans, cls_inds, obj_ids if with_ids else None, stuff_area_limit=200, names=names)
    assert list(got) == names
    seen = set()
    for n in names:
        want = z[f"s{seed}_{int(with_ids)}_{n}"]
        assert got[n].dtype == np.uint8 and got[n].shape == want.shape
        np.testing.assert_array_equal(got[n], want)
        seen |= set(np.unique(want[:, :, 0]).tolist())
    assert 255 in seen and any(c > 10 for c in seen) and any(c <= 10 for c in seen)     # every branch was exercised


def test_duplicate_object_ids_get_fresh_ids():
    seg = np.full((8, 8), 11, np.uint8)
    pan = np.zeros((8, 8), np.uint8)
    pan[0:2], pan[2:4], pan[4:6] = 11, 12, 13
    out = harness.get_unified_pan_result([seg], [pan], [np.array([1, 1, 1])], [np.array([4, 4, 4], np.int32)],
                                         stuff_area_limit=1, names=["a"])["a"]
    # the last duplicate keeps id 4 (+1), the earlier ones get 100, 101 handed out back to front (+1)
    assert [int(out[r, 0, 2]) for r in (0, 2, 4)] == [102, 101, 5]
    assert [int(out[r, 0, 1]) for r in (0, 2, 4)] == [1, 2, 3]


class _FakeDetector(torch.nn.Module):
    def forward(self, img, img_meta, return_loss=True, rescale=None, ref_img=None):
        assert return_loss is False and rescale is True and ref_img is not None
        h, w = img[0].shape[-2:]
        k = int(img_meta[0][0]["iid"] % 10)
        return {"fcn_outputs": torch.full((1, h, w), k, dtype=torch.long), "panoptic_outputs": torch.full((1, h, w), 300 + k),
                "panoptic_cls_inds": torch.tensor([1, 2]), "panoptic_cls_prob": torch.tensor([0.9, 0.95]),
                "panoptic_det_obj_ids": torch.tensor([0, k])}


def test_single_gpu_test_result_layout():
    loader = [dict(img=[torch.zeros(1, 3, 4, 6)], ref_img=[torch.zeros(1, 3, 4, 6)],
                   img_meta=[[dict(iid=10001 + i, filename=f"val/x/f{i}.png")]]) for i in range(3)]
    r = harness.single_gpu_test(_FakeDetector(), loader)
    assert r["all_names"] == ["f0.png", "f1.png", "f2.png"]
    assert r["all_ssegs"][1].dtype == np.uint8 and r["all_ssegs"][1].shape == (4, 6) and r["all_ssegs"][2][0, 0] == 3
    assert r["all_panos"][0][0, 0] == (301 % 256)                        # the harness casts to uint8 (tools/test_vpq.py:44-46)
    assert r["all_pano_obj_ids"][2].tolist() == [0, 3] and r["all_pano_cls_inds"][0].tolist() == [1, 2]


def test_host_pools_restore_the_callers_settings():
    """parallel.host_pools (used by single_gpu_test / clip_gpu_test) sizes the host thread pools for the duration of the loop only."""
    import torch
    from slotvps_amd import parallel
    before = torch.get_num_threads()
    torch.set_num_threads(max(1, before))
    with parallel.host_pools() as hp:
        assert hp.info is not None and hp.info["threads"] <= max(1, before)
    assert torch.get_num_threads() == before
