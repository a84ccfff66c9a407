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


def _compare_with_oracle(cuda, logits, masks, size, cfg=CFG):
    import torch
    from slotvps_amd.postprocess import PostProcessPanopticInstances
    pp = PostProcessPanopticInstances(**cfg)
    res = pp.forward_tensors(torch.from_numpy(logits).to(cuda), torch.from_numpy(masks).to(cuda), size, materialize_masks=True)
    want = po.postprocess(logits, masks, size, threshold=cfg["threshold"], pixel_threshold=cfg["pixel_threshold"],
                          fraction_threshold=cfg["fraction_threshold"])
    np.testing.assert_array_equal(res.slot_index.cpu().numpy(), want["slot_index"])
    np.testing.assert_array_equal(res.labels.cpu().numpy(), want["labels"])
    assert res.area == want["area"]
    got_masks = res.masks.cpu().numpy()
    assert got_masks.shape == want["masks"].shape and np.abs(got_masks - want["masks"]).max() < 2e-6
    ids, cls_inds, _ = pp.panoptic_ids(res)
    want_ids, want_cls, _ = po.panoptic_relabel(want["masks"], want["labels"])
    np.testing.assert_array_equal(ids.cpu().numpy().astype(np.int64), want_ids)
    np.testing.assert_array_equal(cls_inds.numpy(), want_cls)
    return res, want


@pytest.mark.parametrize("seed", range(30, 40))
def test_postprocess_random_cases_match_oracle(cuda, seed):
    """Varying slot counts, kept-slot counts and (non multiple-of-anything) sizes; the oracle is pinned by the
    reference's own outputs (tests/test_post_oracle_golden.py)."""
    rng = np.random.default_rng(seed)
    L = int(rng.choice([20, 50, 100, 200]))
    h, w = int(rng.integers(5, 40)), int(rng.integers(5, 70))
    nk = int(rng.integers(7, min(L, 40)))
    logits, masks = synth.make_post_case(seed, L, h, w, 20, nk)
    _compare_with_oracle(cuda, logits, masks, (4 * h, 4 * w))


def test_postprocess_edge_cases(cuda):
    import torch
    from slotvps_amd.postprocess import PostProcessPanopticInstances
    L, h, w = 12, 8, 16
    base = np.full((L, 20), -4.0, np.float32)
    base[:, 19] = 6.0                                                     # everything "no object" ...
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    blob = lambda cy, cx, s, amp: (amp * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s)) - 1.0).astype(np.float32)
    masks = np.full((L, h, w), -3.0, np.float32)

    # (1) no slot passes the score threshold: the reference's mask_removal crashes (np.stack of nothing, :652)
    pp = PostProcessPanopticInstances(**CFG)
    with pytest.raises(ValueError):
        pp.forward_tensors(torch.from_numpy(base).to(cuda), torch.from_numpy(masks).to(cuda), (4 * h, 4 * w))

    # (2) stuff only, two slots of the same stuff class (merged ids), one class owning no pixel at all
    lg, mk = base.copy(), masks.copy()
    for s, c in [(0, 3), (1, 3), (2, 7)]:
        lg[s] = -4.0
        lg[s, c] = 8.0
    mk[0], mk[1] = blob(2, 3, 4.0, 6.0), blob(6, 12, 4.0, 6.0)
    _compare_with_oracle(cuda, lg, mk, (4 * h, 4 * w))

    # (3) things only: identical twins of one class (the weaker one is removed by the overlap rule), an instance of
    #     another class on top of them (kept: the rule is per class), an instance that never reaches 0.4 anywhere
    lg, mk = base.copy(), masks.copy()
    for s, c, sc in [(3, 13, 9.0), (4, 13, 8.0), (5, 15, 8.5), (6, 16, 8.2), (7, 2, 8.0)]:
        lg[s] = -4.0
        lg[s, c] = sc
    mk[3] = mk[4] = blob(4, 8, 2.0, 8.0)
    mk[5] = blob(4, 8, 1.5, 9.0)
    mk[6] = np.full((h, w), -2.9, np.float32)                            # flat, below every blob: no candidate pixel
    mk[7] = np.zeros((h, w), np.float32)                                  # a stuff floor so that the argmax has a background
    res, want = _compare_with_oracle(cuda, lg, mk, (4 * h, 4 * w))
    assert 4 not in res.slot_index.tolist() and 6 not in res.slot_index.tolist() and 3 in res.slot_index.tolist()

    # (4) exact ties between two kept slots: the first in the reference's order wins every pixel
    lg, mk = base.copy(), masks.copy()
    for s, c, sc in [(8, 1, 9.0), (9, 4, 8.0)]:
        lg[s] = -4.0
        lg[s, c] = sc
    mk[8] = mk[9] = blob(3, 5, 3.0, 4.0)
    res, want = _compare_with_oracle(cuda, lg, mk, (4 * h, 4 * w))
    assert res.slot_index.tolist() == [8]                                 # slot 9 owns no pixel -> area 0 -> filtered


def _clip_vs_frames(cuda, cases, size, rounds=None, device_decisions=True, cfg=CFG):
    """The frames of `cases` (same L, h, w) through forward_clip / panoptic_ids_clip against forward_tensors / panoptic_ids frame by
    frame: kept slots, labels, areas and the id maps must be identical."""
    import torch
    from slotvps_amd.postprocess import PostProcessPanopticInstances
    pp = PostProcessPanopticInstances(**cfg)
    pp.device_decisions = device_decisions
    if rounds is not None:
        pp.clip_rounds = rounds
    lg = torch.from_numpy(np.stack([c[0] for c in cases])).to(cuda)
    mk = torch.from_numpy(np.stack([c[1] for c in cases])).to(cuda)
    res = pp.forward_clip(lg, mk, size)
    pans = pp.panoptic_ids_clip(res)
    ref = PostProcessPanopticInstances(**cfg)
    for t in range(len(cases)):
        want = ref.forward_tensors(lg[t], mk[t], size)
        w_ids, w_cls, w_prob = ref.panoptic_ids(want)
        np.testing.assert_array_equal(res[t].slot_index.cpu().numpy(), want.slot_index.cpu().numpy())
        np.testing.assert_array_equal(res[t].slot_index_host, want.slot_index.cpu().numpy())
        np.testing.assert_array_equal(res[t].labels_host, want.labels.cpu().numpy())
        np.testing.assert_array_equal(res[t].probs_host, want.probs.cpu().numpy())
        assert res[t].area == want.area
        ids, cls, prob = pans[t]
        assert torch.equal(ids.view(size[0], size[1]), w_ids)
        np.testing.assert_array_equal(cls.numpy(), w_cls.numpy())
        np.testing.assert_array_equal(prob.numpy(), w_prob.cpu().numpy())
    return res


@pytest.mark.parametrize("device_decisions", [True, False])
def test_clip_postprocess_equals_frame_postprocess(cuda, device_decisions):
    """K6c (decisions on the device, 4 x 4 output blocks, one copy per clip) and the lock-step host path against the per-frame path,
    which the tests above pin to the reference's own outputs and the oracle: ragged sizes (w not a multiple of the 64-cell block),
    20 ... 200 slots, 7 ... 39 kept."""
    for seed in range(50, 58):
        rng = np.random.default_rng(seed)
        L = int(rng.choice([20, 50, 100, 200]))
        h, w = int(rng.integers(5, 40)), int(rng.integers(5, 90))
        T = int(rng.integers(1, 6))
        cases = [synth.make_post_case(1000 * seed + t, L, h, w, 20, int(rng.integers(7, min(L, 40)))) for t in range(T)]
        _clip_vs_frames(cuda, cases, (4 * h, 4 * w), device_decisions=device_decisions)
    # the reference's own fixture inputs, as one-frame clips
    z = np.load(os.path.join(GOLDEN, "postprocess.npz"))
    for tag in "abc":
        seed, L, h, w, nk = (int(x) for x in z[f"{tag}_meta"])
        res = _clip_vs_frames(cuda, [synth.make_post_case(seed, L, h, w, 20, nk)], (4 * h, 4 * w), device_decisions=device_decisions)
        np.testing.assert_array_equal(res[0].slot_index_host, z[f"{tag}_slot_index"])
        np.testing.assert_array_equal(res[0].labels_host, z[f"{tag}_labels"])


def test_clip_postprocess_more_rounds_than_enqueued(cuda):
    """The small-area loop has a data-dependent trip count: with ONE speculative round enqueued, frames that need more must be
    continued by the host - same results; and at the full size (1024 x 2048 from 256 x 512)."""
    cases = [synth.make_post_case(77 + t, 100, 20, 36, 20, 30) for t in range(4)]
    res = _clip_vs_frames(cuda, cases, (80, 144), rounds=1)
    assert max(r.rounds for r in res) >= 1                      # at least one frame went round the small-area loop
    cases = [synth.make_post_case(21 + t, 100, 256, 512, 20, 30) for t in range(2)]
    _clip_vs_frames(cuda, cases, (1024, 2048))


def test_clip_postprocess_edge_cases(cuda):
    """Stuff only with duplicated classes (de-duplication table, then the identity pass), things only with twins, nothing kept by the
    score filter (must raise like the per-frame path)."""
    import torch
    from slotvps_amd.postprocess import PostProcessPanopticInstances
    L, h, w = 12, 8, 16
    base = np.full((L, 20), -4.0, np.float32)
    base[:, 19] = 6.0
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    blob = lambda cy, cx, s, amp: (amp * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s)) - 1.0).astype(np.float32)
    masks = np.full((L, h, w), -3.0, np.float32)
    pp = PostProcessPanopticInstances(**CFG)
    with pytest.raises(ValueError):
        pp.forward_clip(torch.from_numpy(base[None]).to(cuda), torch.from_numpy(masks[None]).to(cuda), (4 * h, 4 * w))
    lg, mk = base.copy(), masks.copy()
    for s, c in [(0, 3), (1, 3), (2, 7)]:
        lg[s] = -4.0
        lg[s, c] = 8.0
    mk[0], mk[1] = blob(2, 3, 4.0, 6.0), blob(6, 12, 4.0, 6.0)
    lg2, mk2 = base.copy(), masks.copy()
    for s, c in [(0, 12), (1, 12), (2, 15), (3, 4)]:
        lg2[s] = -4.0
        lg2[s, c] = 8.0 - 0.1 * s
    mk2[0], mk2[1], mk2[2], mk2[3] = blob(3, 4, 2.0, 8.0), blob(3, 4, 2.0, 7.5), blob(5, 11, 1.5, 8.0), blob(4, 8, 6.0, 3.0)
    for dd in (True, False):
        _clip_vs_frames(cuda, [(lg, mk), (lg2, mk2), (lg, mk)], (4 * h, 4 * w), device_decisions=dd)


def test_clip_postprocess_equal_scores(cuda):
    """Two kept slots with bit-identical class logits have equal scores; the reference's order among them is an accident of numpy's
    unstable argsort (AVX-512 and scalar builds differ). Every path here - per frame, clip on the host, clip on the device, and the
    oracle - takes the scalar path's order (ties in descending slot order): they must agree."""
    lg, mk = synth.make_post_case(91, 50, 12, 20, 20, 12)
    order = np.argsort(-lg.max(-1))
    lg[order[3]] = lg[order[2]]                                  # a tie among the kept slots
    for dd in (True, False):
        _clip_vs_frames(cuda, [(lg, mk), synth.make_post_case(92, 50, 12, 20, 20, 9)], (48, 80), device_decisions=dd)
    _compare_with_oracle(cuda, lg, mk, (48, 80))


def test_clip_postprocess_many_kept_slots(cuda):
    """More kept slots than the decision kernel holds pair counts for in LDS (K > 120: the pair table is read from global memory) - the
    VIPER geometry's worst case (200 slots)."""
    cases = [synth.make_post_case(300 + t, 200, 17, 30, 20, nk) for t, nk in enumerate((150, 199, 121))]
    res = _clip_vs_frames(cuda, cases, (68, 120))
    assert max(len(r._thing) for r in res) > 120


@pytest.mark.parametrize("overflow_frame", [0, 2])
def test_clip_postprocess_decode_cap(cuda, overflow_frame):
    """The clip path decodes the first `clip_decode_cap` slots of every frame's score order before the host knows K. A frame that keeps
    more (here: the first / the LAST frame of the clip, whose missing rows would lie past the end of the decoded buffer) is skipped by the
    kernels and the clip runs again with all rows: same results as without a cap; frames within the cap never take the second pass."""
    from slotvps_amd.postprocess import PostProcessPanopticInstances
    nk = [9, 9, 9]
    nk[overflow_frame] = 30
    cases = [synth.make_post_case(400 + t, 100, 12, 24, 20, k) for t, k in enumerate(nk)]
    saved = PostProcessPanopticInstances.clip_decode_cap
    try:
        PostProcessPanopticInstances.clip_decode_cap = 16          # 30 kept > 16: the rerun path
        res = _clip_vs_frames(cuda, cases, (48, 96))
        assert res[0]._row_stride == 100
        PostProcessPanopticInstances.clip_decode_cap = 32          # all frames within the cap: one pass on 32 rows
        res = _clip_vs_frames(cuda, cases, (48, 96))
        assert res[0]._row_stride == 32
        PostProcessPanopticInstances.clip_decode_cap = None        # no cap: all L rows
        res = _clip_vs_frames(cuda, cases, (48, 96))
        assert res[0]._row_stride == 100
        # the overflow is STICKY per instance - the clip that overflows runs twice, the next clip of the same model starts with
        # a cap that covers what was kept (next power of two, at most L) and takes one pass
        import torch
        PostProcessPanopticInstances.clip_decode_cap = 16
        pp = PostProcessPanopticInstances(**CFG)
        lg = torch.from_numpy(np.stack([c[0] for c in cases])).to(cuda)
        mk = torch.from_numpy(np.stack([c[1] for c in cases])).to(cuda)
        first = pp.forward_clip(lg, mk, (48, 96))
        assert first[0]._row_stride == 100 and pp.clip_decode_cap == 32 and PostProcessPanopticInstances.clip_decode_cap == 16
        second = pp.forward_clip(lg, mk, (48, 96))
        assert second[0]._row_stride == 32
        for a_, b_ in zip(first, second):
            np.testing.assert_array_equal(a_.slot_index_host, b_.slot_index_host)
            np.testing.assert_array_equal(a_.labels_host, b_.labels_host)
    finally:
        PostProcessPanopticInstances.clip_decode_cap = saved
