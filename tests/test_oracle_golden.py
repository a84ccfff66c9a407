"""Pins the CPU oracle against golden vectors captured from the reference's own modules
(tests/golden/make_golden.py). CPU only; float32 arithmetic like the reference, tolerances cover
BLAS summation-order noise only."""
import os

import numpy as np
import pytest

import synth
from util import orc, GOLDEN

F32 = np.float32


def _load(name):
    path = os.path.join(GOLDEN, name)
    assert os.path.exists(path), f"missing fixture {path} (run tests/golden/make_golden.py in the build container)"
    return np.load(path)


@pytest.mark.parametrize("H,W", [(2, 4), (16, 32), (33, 65), (34, 60)])
def test_pos_embed_sine(H, W):
    ref = _load("pos_embed_sine.npz")[f"pos_{H}x{W}"]
    got = orc.pos_embed_sine(H, W, 256)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 2e-6


@pytest.mark.parametrize("tag", ["L100_16x32", "L37_9x13", "L200_6x10"])
def test_retriever(tag):
    z = _load("retriever.npz")
    L, H, W, seed = (int(x) for x in z[f"{tag}_meta"])
    params = synth.make_params(synth.retriever_shapes(""), seed)
    rng = np.random.default_rng(seed + 1000)
    slots = rng.standard_normal((1, L, 256)).astype(F32)[0]
    feat = synth.smooth_features(rng, 256, H, W).reshape(256, H * W).T
    got = orc.retriever(slots, feat, orc.pos_embed_sine(H, W), params, "")
    err = np.abs(got - z[f"{tag}_out"]).max()
    assert err < 2e-5, err


@pytest.mark.parametrize("tag,act", [("N200_relu", "relu"), ("N150_gelu", "gelu")])
def test_temporal_head(tag, act):
    z = _load("temporal_head.npz")
    N, ff, seed = (int(x) for x in z[f"{tag}_meta"])
    params = synth.make_params(synth.temporal_shapes("", ff), seed)
    S = np.random.default_rng(seed + 1000).standard_normal((N, 256)).astype(F32)
    # slot<->slot logits reach +-80: fp32 summation-order noise is amplified by exp() to ~5e-5
    inner = orc.slots_retriever(S, S, params, "inst_interact.")
    assert np.abs(inner - z[f"{tag}_inner"]).max() < 2e-4
    got = orc.temporal_head(S, params, "", act)
    assert np.abs(got - z[f"{tag}_out"]).max() < 2e-4


@pytest.mark.parametrize("tag", ["T2_64x128", "T3_64x64"])
def test_whole_head_and_decode(tag):
    z = _load("head_small.npz")
    T, H, W, L, seed = (int(x) for x in z[f"{tag}_meta"])
    params = synth.make_params(synth.head_shapes(), seed)
    feats = synth.make_clip_features(seed + 1, T, H, W)
    slots = synth.make_slots(seed + 2, L)
    pos = [orc.pos_embed_sine(h, w) for (h, w) in synth.level_sizes(H, W)]
    logits, embeds, fused = orc.head_forward(feats, slots, pos, params)
    for t in range(T):
        el = np.abs(np.stack(logits[t]) - z[f"{tag}_logits_{t}"]).max()
        ee = np.abs(np.stack(embeds[t]) - z[f"{tag}_embeds_{t}"]).max()
        ef = max(np.abs(fused[t][3] - z[f"{tag}_fused3_{t}"]).max(), np.abs(fused[t][0] - z[f"{tag}_fused0_{t}"]).max())
        # Seven chained stages: the reference's own fp32 result moves by up to ~6e-4 on the last-stage
        # embeddings when only the summation order changes (float64 oracle vs reference: 2e-5 at
        # stage 0 growing to 6.5e-4 at stage 6, see DESIGN.md "numerical noise floor"), so 1e-3 is
        # the tightest meaningful bound for the chain; the first stage is held to 5e-5.
        assert el < 1e-3 and ee < 1e-3 and ef < 2e-5, (t, el, ee, ef)
        assert np.abs(embeds[t][0] - z[f"{tag}_embeds_{t}"][0]).max() < 5e-5
    w, b, mu, var = z[f"{tag}_bn"]
    scale, shift = orc.bn_eval_affine(w, b, mu, var)
    fg = z[f"{tag}_fg"]
    fgs, fgb = orc.bn_eval_affine(fg[0], fg[1], fg[2], fg[3])
    # decode from the REFERENCE's own fused map / embedding so this pins a8 alone
    m = orc.mask_decode(z[f"{tag}_fused3_{T - 1}"], z[f"{tag}_embeds_{T - 1}"][-1], scale, shift, fgs, fgb)
    assert np.abs(m - z[f"{tag}_mask"]).max() < 2e-6
    # integer parity on the same inputs wherever the decision margin exceeds the float noise
    ref = z[f"{tag}_mask"]
    srt = np.sort(ref, axis=0)
    decided = (srt[-1] - srt[-2]) > 1e-5
    np.testing.assert_array_equal(orc.slot_argmax(m)[decided], orc.slot_argmax(ref)[decided])


def test_round_bf16_matches_torch():
    import torch
    x = np.random.default_rng(0).standard_normal(100000).astype(F32) * 37.0
    x[:5] = [0.0, -0.0, 1.0039062, 3.3895314e38, 1e-40]
    want = torch.from_numpy(x).to(torch.bfloat16).float().numpy()
    np.testing.assert_array_equal(orc.round_bf16(x), want)


def test_column_sum_invariant():
    """softmax columns sum to 1 over slots => sum_l pre[l] == sum_p v[p]."""
    rng = np.random.default_rng(5)
    q, k, v = rng.standard_normal((100, 256)), rng.standard_normal((700, 256)), rng.standard_normal((700, 256))
    _, pre = orc.retriever_core(q, k, v, np.ones(256), np.zeros(256), return_pre=True)
    assert np.abs(pre.sum(0) - v.sum(0)).max() < 1e-9
