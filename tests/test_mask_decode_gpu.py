"""K2 parity: HIP slot->mask decode vs the CPU oracle on the same bf16 feature map.
Tolerance 1e-4 absolute on the mask logits (north star), bit-exact slot argmax wherever the
oracle's top-2 margin exceeds that tolerance (ties inside the float tolerance are not decidable)."""
import numpy as np
import pytest

from util import orc, to_bf16_t, bf16_t_to_np

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _run(cuda, T, L, HW, seed, default_bn):
    import torch
    from slotvps_amd import ops
    rng = np.random.default_rng(seed)
    feat = (rng.standard_normal((T, HW, 256)) * 1.5).astype(np.float32)
    emb = np.maximum(rng.standard_normal((T, L, 256)), 0).astype(np.float32) * 2.0   # post-ReLU embeddings
    if default_bn:   # freshly initialised model: running stats (0, 1), feat_bn weight 1, fg_bn weight 0.1
        scale, shift = orc.bn_eval_affine(np.ones(256), np.zeros(256), np.zeros(256), np.ones(256))
        fgs, fgb = orc.bn_eval_affine(np.float64(0.1), np.float64(0.0), np.float64(0.0), np.float64(1.0))
    else:
        scale, shift = orc.bn_eval_affine(rng.uniform(0.5, 1.5, 256), 0.2 * rng.standard_normal(256),
                                          0.3 * rng.standard_normal(256), rng.uniform(0.5, 2.0, 256))
        fgs, fgb = orc.bn_eval_affine(np.float64(0.37), np.float64(-0.11), np.float64(0.8), np.float64(2.5))
    tf = to_bf16_t(feat, cuda)
    te = torch.from_numpy(emb).to(cuda)
    out, amax = ops.mask_decode(tf, te, torch.from_numpy(scale.astype(np.float32)).to(cuda),
                                torch.from_numpy(shift.astype(np.float32)).to(cuda), float(fgs), float(fgb),
                                want_argmax=True)
    out2 = ops.mask_decode(tf, te, torch.from_numpy(scale.astype(np.float32)).to(cuda),
                           torch.from_numpy(shift.astype(np.float32)).to(cuda), float(fgs), float(fgb))
    torch.cuda.synchronize()
    assert torch.equal(out, out2), "argmax variant changed the logits"
    if HW % 4 == 0:                                  # argmax-only mode (out == NULL, the fast kernel without its logit stores)
        none, amax_only = ops.mask_decode(tf, te, torch.from_numpy(scale.astype(np.float32)).to(cuda),
                                          torch.from_numpy(shift.astype(np.float32)).to(cuda), float(fgs), float(fgb),
                                          want_argmax=True, want_logits=False)
        torch.cuda.synchronize()
        assert none is None
        # the argmax-only kernel orders the slots by sgn(fg_scale) * (e . g) without the per-pixel norm: identical to the full
        # mode's argmax except at pixels whose two candidates' fp32 logits coincide (no decision at fp32 resolution)
        diff = (amax_only != amax)
        if bool(diff.any()):
            a = torch.gather(out, 1, amax.long().unsqueeze(1)).squeeze(1)
            b = torch.gather(out, 1, amax_only.long().unsqueeze(1)).squeeze(1)
            gap = ((a - b).abs() / a.abs().clamp_min(1e-30))[diff]
            assert float(gap.max()) <= 2.5e-7, f"argmax-only mode differs from the full mode at a decided pixel (relative gap {float(gap.max())})"
            assert float(diff.float().mean()) < 1e-3
    out, amax = out.cpu().numpy(), amax.cpu().numpy()
    ff = bf16_t_to_np(tf).astype(np.float64)
    s32, h32 = scale.astype(np.float32).astype(np.float64), shift.astype(np.float32).astype(np.float64)
    worst = 0.0
    for t in range(T):
        ref = orc.mask_decode(ff[t], emb[t].astype(np.float64), s32, h32, float(np.float32(fgs)), float(np.float32(fgb)))
        worst = max(worst, float(np.abs(out[t] - ref).max()))
        # integer parity: kernel argmax == argmax of the kernel's own logits (first max wins) ...
        np.testing.assert_array_equal(amax[t], orc.slot_argmax(out[t]))
        # ... and == the oracle's argmax wherever the oracle's decision margin is above tolerance
        srt = np.sort(ref, axis=0)
        decided = (srt[-1] - srt[-2]) > 2 * TOL if L > 1 else np.ones(HW, bool)
        np.testing.assert_array_equal(amax[t][decided], orc.slot_argmax(ref)[decided])
    return worst


@pytest.mark.parametrize("T,L,HW,default_bn", [
    (1, 100, 512, True),
    (2, 100, 2145, False),     # ragged tile
    (1, 1, 40, False),
    (2, 200, 2040, False),     # VIPER slots (8-wave form of the fast kernel)
    (1, 256, 8192, True),      # the most slots the fast kernel takes, several tiles per workgroup
    (3, 129, 2148, False),     # one slot past four waves, ragged last tile
    (1, 200, 2050, False),     # more than 128 slots with HW % 4 != 0: first-generation kernel
    (1, 128, 8192, True),
    (20, 100, 6144, False),    # several tiles per workgroup (counted DMA / store ring, deferred argmax), fast path
    (20, 100, 6148, False),    # the same with a ragged last tile
    (3, 100, 2050, False),     # HW % 4 != 0: scalar-store kernel
    (40, 100, 768, False),     # two tiles per workgroup: shorter than the DMA ring's prologue (skewed loop: first + last iteration only)
    (40, 100, 1152, True),     # three tiles per workgroup
    (40, 100, 1540, False),    # five tiles per workgroup, last chunk short and ragged
    (24, 200, 1540, False),    # the same for the 8-wave form
])
def test_mask_decode_matches_oracle(cuda, T, L, HW, default_bn):
    worst = _run(cuda, T, L, HW, seed=L + HW, default_bn=default_bn)
    assert worst <= TOL, f"mask-logit max abs err {worst:.3e}"


def test_mask_decode_full_size_properties(cuda):
    """BASELINE size: one 256x512 frame, L=100. Linearity in the slot embeddings (decode(a e1 + b e2) =
    a decode(e1) + b decode(e2) when fg_shift = 0) and |logit| <= fg_scale * ||e|| (unit-norm features)."""
    import torch
    from slotvps_amd import ops
    T, L, HW = 1, 100, 256 * 512
    g = torch.Generator(device=cuda).manual_seed(5)
    feat = torch.randn((T, HW, 256), generator=g, device=cuda).to(torch.bfloat16)
    e1 = torch.relu(torch.randn((T, L, 256), generator=g, device=cuda))
    e2 = torch.relu(torch.randn((T, L, 256), generator=g, device=cuda))
    sc = torch.ones(256, device=cuda)
    sh = torch.zeros(256, device=cuda)
    d1 = ops.mask_decode(feat, e1, sc, sh, 0.1, 0.0)
    d2 = ops.mask_decode(feat, e2, sc, sh, 0.1, 0.0)
    d12 = ops.mask_decode(feat, 2.0 * e1 + 0.5 * e2, sc, sh, 0.1, 0.0)
    torch.cuda.synchronize()
    assert (d12 - (2.0 * d1 + 0.5 * d2)).abs().max().item() < 1e-4
    bound = 0.1 * e1.norm(dim=2)            # [T, L]
    assert (d1.abs().amax(dim=2) <= bound * (1 + 1e-5) + 1e-6).all()


@pytest.mark.parametrize("fg_scale", [-0.37, 0.0])
def test_argmax_only_mode_follows_the_sign_of_the_foreground_scale(cuda, fg_scale):
    """Argmax-only mode skips the per-pixel norm (it cannot change the order of the slots); a negative foreground scale
    reverses the order, a zero scale makes every slot equal (first index wins) - as in the full mode."""
    import torch
    from slotvps_amd import ops
    T, L, HW = 2, 100, 4096
    g = torch.Generator(device=cuda).manual_seed(11)
    feat = torch.randn((T, HW, 256), generator=g, device=cuda).to(torch.bfloat16)
    e = torch.relu(torch.randn((T, L, 256), generator=g, device=cuda))
    sc = torch.rand(256, generator=g, device=cuda) + 0.5
    sh = 0.2 * torch.randn(256, generator=g, device=cuda)
    full, amax = ops.mask_decode(feat, e, sc, sh, fg_scale, 0.05, want_argmax=True)
    _, only = ops.mask_decode(feat, e, sc, sh, fg_scale, 0.05, want_argmax=True, want_logits=False)
    torch.cuda.synchronize()
    if fg_scale == 0.0:
        assert int(only.max()) == 0 and int(amax.max()) == 0
        return
    diff = only != amax
    a = torch.gather(full, 1, amax.long().unsqueeze(1)).squeeze(1)
    b = torch.gather(full, 1, only.long().unsqueeze(1)).squeeze(1)
    assert float(diff.float().mean()) < 1e-3
    if bool(diff.any()):
        assert float(((a - b).abs() / a.abs().clamp_min(1e-30))[diff].max()) <= 2.5e-7


@pytest.mark.parametrize("T,L,HW", [(2, 100, 2048), (1, 200, 2040), (2, 100, 2145), (20, 100, 6144)])
def test_mask_decode_fp16_map(cuda, T, L, HW):
    """An fp16 fused map (MultiScaleDynamicMaskHead.map_dtype = "fp16"): fp16 MFMAs, the slot operand as fp16 hi + lo; against the
    oracle on the same fp16 values <= 1e-4, argmax-only mode included."""
    import torch
    from slotvps_amd import ops
    rng = np.random.default_rng(L + HW)
    feat = rng.standard_normal((T, HW, 256)).astype(np.float32)
    emb = np.maximum(rng.standard_normal((T, L, 256)), 0).astype(np.float32)
    scale, shift = orc.bn_eval_affine(rng.uniform(0.5, 1.5, 256), 0.2 * rng.standard_normal(256), 0.3 * rng.standard_normal(256), rng.uniform(0.5, 2.0, 256))
    fgs, fgb = orc.bn_eval_affine(np.float64(0.37), np.float64(-0.11), np.float64(0.8), np.float64(2.5))
    tf = torch.from_numpy(feat).to(cuda).to(torch.float16).contiguous()
    te = torch.from_numpy(emb).to(cuda)
    ts, th = torch.from_numpy(scale.astype(np.float32)).to(cuda), torch.from_numpy(shift.astype(np.float32)).to(cuda)
    out, amax = ops.mask_decode(tf, te, ts, th, float(fgs), float(fgb), want_argmax=True)
    torch.cuda.synchronize()
    ff = tf.float().cpu().numpy().astype(np.float64)
    s32, h32 = scale.astype(np.float32).astype(np.float64), shift.astype(np.float32).astype(np.float64)
    worst = 0.0
    for t in range(T):
        ref = orc.mask_decode(ff[t], emb[t].astype(np.float64), s32, h32, float(np.float32(fgs)), float(np.float32(fgb)))
        worst = max(worst, float(np.abs(out[t].cpu().numpy() - ref).max()))
        np.testing.assert_array_equal(amax[t].cpu().numpy(), orc.slot_argmax(out[t].cpu().numpy()))
    print(f"K2 fp16 map T={T} L={L} HW={HW}: {worst:.2e}")
    assert worst <= TOL, worst
    if HW % 4 == 0:
        none, amax_only = ops.mask_decode(tf, te, ts, th, float(fgs), float(fgb), want_argmax=True, want_logits=False)
        assert none is None and float((amax_only != amax).float().mean()) < 1e-3
