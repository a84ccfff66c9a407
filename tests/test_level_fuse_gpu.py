"""K4 parity: HIP level fusion (bilinear x2 + concat + shared 1x1 conv, dynamic_mask_head.py:171-188)
vs the CPU oracle under the bf16 storage policy. bf16 outputs: every element within one bf16 ulp
(+1e-5 absolute) of the oracle, rounding flips rare."""
import numpy as np
import pytest

import synth
from util import orc, to_bf16_t, bf16_t_to_np

pytestmark = pytest.mark.gpu


def _ulp_bf16(x):
    return 2.0 ** (np.floor(np.log2(np.maximum(np.abs(x), 1e-30))) - 7)


@pytest.mark.parametrize("T,H,W,level0,nchw", [
    (2, 8, 16, True, True),
    (1, 8, 16, True, False),
    (2, 16, 32, False, True),
    (1, 16, 32, False, False),
    (2, 68, 120, False, True),      # VIPER level 1 from a 34x60 level 0: rows not tile aligned, ragged end
    (1, 34, 60, True, True),
    (3, 64, 128, False, True),
    (9, 32, 64, False, True),       # several tiles per workgroup on the staged-tap fast path (W % 32 == 0)
    (9, 32, 64, False, False),
])
def test_level_fuse_matches_oracle(cuda, T, H, W, level0, nchw):
    import torch
    from slotvps_amd import ops
    seed = 31 * H + W + T
    rng = np.random.default_rng(seed)
    params = synth.make_params({"conv_trans.conv.weight": (256, 384, 1, 1), "conv_trans.conv.bias": (256,)}, seed)
    wc = params["conv_trans.conv.weight"].reshape(256, 384)
    bc = params["conv_trans.conv.bias"]
    cur = np.stack([synth.smooth_features(rng, 128, H, W) for _ in range(T)])
    prev_pm = None
    if not level0:
        prev_pm = np.stack([synth.smooth_features(rng, 256, H // 2, W // 2).reshape(256, -1).T * 1.5 for _ in range(T)])
    st = orc.Storage.bf16_policy()
    t_prev = to_bf16_t(prev_pm, cuda) if prev_pm is not None else None
    if nchw:
        t_cur = torch.from_numpy(cur).to(cuda)
        cur_seen = cur
    else:
        t_cur = to_bf16_t(cur.reshape(T, 128, H * W).transpose(0, 2, 1), cuda)
        cur_seen = bf16_t_to_np(t_cur).transpose(0, 2, 1).reshape(T, 128, H, W)
    out = ops.level_fuse(t_cur, t_prev, torch.from_numpy(wc).to(cuda).to(torch.bfloat16).contiguous(),
                         torch.from_numpy(bc).to(cuda), H, W)
    torch.cuda.synchronize()
    out = bf16_t_to_np(out)
    flips = total = 0
    worst = 0.0
    for t in range(T):
        prev = None
        if not level0:
            prev = bf16_t_to_np(t_prev)[t].T.reshape(256, H // 2, W // 2)
        ref = orc.fuse_level(cur_seen[t].astype(np.float32), prev, wc, bc, st)
        d = np.abs(out[t] - ref)
        ulp = _ulp_bf16(np.maximum(np.abs(out[t]), np.abs(ref)))
        ij = np.unravel_index(np.argmax(d - ulp), d.shape)
        assert (d <= ulp * 1.001 + 1e-5).all(), f"more than one bf16 ulp off at {ij}: got {out[t][ij]!r} ref {ref[ij]!r}"
        flips += int((d > 0).sum())
        total += d.size
        worst = max(worst, d.max())
    print(f"\n[level_fuse T{T} {H}x{W} level0={level0} nchw={nchw}] max |d| {worst:.3e}, flipped {100 * flips / total:.3f}%")
    assert flips / total < 0.01


def test_level_fuse_full_size_properties(cuda):
    """BASELINE size (T=5, 256x512 from a 128x256 level): zero inputs give the bias everywhere; a constant previous level
    and a constant incoming map give one constant output vector (the bilinear weights sum to 1, borders included); frames
    are independent of their batch."""
    import torch
    from slotvps_amd import ops
    T, H, W = 5, 256, 512
    g = torch.Generator(device=cuda).manual_seed(9)
    wc = (torch.randn((256, 384), generator=g, device=cuda) / 20).to(torch.bfloat16)
    bc = torch.randn(256, generator=g, device=cuda)
    cur0 = torch.zeros((T, 128, H, W), device=cuda)
    prev0 = torch.zeros((T, H * W // 4, 256), device=cuda, dtype=torch.bfloat16)
    out = ops.level_fuse(cur0, prev0, wc, bc, H, W)
    assert torch.equal(out, bc.to(torch.bfloat16).expand(T, H * W, 256))
    cvec = torch.randn(128, generator=g, device=cuda).to(torch.bfloat16).float()
    pvec = torch.randn(256, generator=g, device=cuda).to(torch.bfloat16)
    cur = cvec.view(1, 128, 1, 1).expand(T, 128, H, W).contiguous()
    prev = pvec.expand(T, H * W // 4, 256).contiguous()
    out = ops.level_fuse(cur, prev, wc, bc, H, W)
    want = (torch.cat([pvec.float(), cvec]) @ wc.float().t() + bc).to(torch.bfloat16)
    d = (out.float() - want.float()).abs().max().item()
    assert d <= 2 * 2.0 ** -5, d                                       # fp32 accumulation order across one bf16 rounding, |x| < 8
    assert torch.equal(out[0], out[4])
    cur_r = torch.randn((T, 128, H, W), generator=g, device=cuda)
    prev_r = torch.randn((T, H * W // 4, 256), generator=g, device=cuda).to(torch.bfloat16)
    full = ops.level_fuse(cur_r, prev_r, wc, bc, H, W)
    one = ops.level_fuse(cur_r[2:3].contiguous(), prev_r[2:3].contiguous(), wc, bc, H, W)
    assert torch.equal(one[0], full[2])


@pytest.mark.parametrize("T,H,W", [(2, 16, 32), (9, 32, 64), (5, 64, 128)])
def test_level_fuse_is_bitwise_reproducible(cuda, T, H, W):
    """Five launches on the same (non-smooth) inputs give the same bytes: a request or store that lands where it should not - the
    wave-specialised fast path issues out-of-range "dropped" requests past the end of a chunk - shows up as a run-to-run difference
    (it did once: a signed overflow in a dropped store's offset made it land in range)."""
    import torch
    from slotvps_amd import ops
    g = torch.Generator(device=cuda).manual_seed(H + W)
    cur = torch.randn((T, 128, H, W), generator=g, device=cuda)
    prev = torch.randn((T, (H // 2) * (W // 2), 256), generator=g, device=cuda).to(torch.bfloat16)
    wc = (torch.randn((256, 384), generator=g, device=cuda) * 0.05).to(torch.bfloat16)
    bc = torch.randn(256, generator=g, device=cuda)
    runs = []
    for _ in range(5):
        o = ops.level_fuse(cur, prev, wc, bc, H, W)
        torch.cuda.synchronize()
        runs.append(o.view(torch.int16).cpu())
        del o
    assert all(torch.equal(runs[0], r) for r in runs[1:])


@pytest.mark.parametrize("T,H,W,level0", [(2, 8, 16, True), (2, 16, 32, False), (2, 68, 120, False), (9, 32, 64, False), (3, 64, 128, False)])
def test_level_fuse_fp16_maps_match_oracle(cuda, T, H, W, level0):
    """map_dtype = "fp16": incoming map, previous level, conv weight and result as fp16 (three more mantissa bits in the same bytes)
    against the oracle under the fp16 storage policy: every element within one fp16 ulp; and 8x closer to the unrounded conv than
    the bf16 form."""
    import torch
    from slotvps_amd import ops
    seed = 31 * H + W + T
    rng = np.random.default_rng(seed)
    params = synth.make_params({"conv_trans.conv.weight": (256, 384, 1, 1), "conv_trans.conv.bias": (256,)}, seed)
    wc = params["conv_trans.conv.weight"].reshape(256, 384)
    bc = params["conv_trans.conv.bias"]
    cur = np.stack([synth.smooth_features(rng, 128, H, W) for _ in range(T)])
    prev_pm = None
    if not level0:
        prev_pm = np.stack([synth.smooth_features(rng, 256, H // 2, W // 2).reshape(256, -1).T * 1.5 for _ in range(T)])
    st = orc.Storage.fused_fp16_policy()
    t_prev16 = torch.from_numpy(prev_pm).to(cuda).to(torch.float16).contiguous() if prev_pm is not None else None
    t_prevbf = to_bf16_t(prev_pm, cuda) if prev_pm is not None else None
    t_cur = torch.from_numpy(cur).to(cuda)
    twc = torch.from_numpy(wc).to(cuda)
    out16 = ops.level_fuse(t_cur, t_prev16, twc.to(torch.float16).contiguous(), torch.from_numpy(bc).to(cuda), H, W)
    outbf = ops.level_fuse(t_cur, t_prevbf, twc.to(torch.bfloat16).contiguous(), torch.from_numpy(bc).to(cuda), H, W)
    again = ops.level_fuse(t_cur, t_prev16, twc.to(torch.float16).contiguous(), torch.from_numpy(bc).to(cuda), H, W)
    torch.cuda.synchronize()
    assert out16.dtype == torch.float16 and torch.equal(out16, again)
    o16, obf = out16.float().cpu().numpy(), bf16_t_to_np(outbf)
    e16 = ebf = 0.0
    for t in range(T):
        prev16 = t_prev16[t].float().cpu().numpy().T.reshape(256, H // 2, W // 2) if not level0 else None
        ref = orc.fuse_level(cur[t].astype(np.float32), prev16, wc, bc, st)
        d = np.abs(o16[t] - ref)
        ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.maximum(np.abs(o16[t]), np.abs(ref)), 2.0 ** -14))) - 10)
        ij = np.unravel_index(np.argmax(d - ulp), d.shape)
        assert (d <= ulp * 1.001 + 1e-5).all(), f"more than one fp16 ulp (+ 1e-5: fp32 summation order near zero) off at {ij}: got {o16[t][ij]!r} ref {ref[ij]!r}"
        assert (d > 0).mean() < 0.02
        exact = orc.fuse_level(cur[t].astype(np.float64), None if level0 else prev_pm[t].T.reshape(256, H // 2, W // 2).astype(np.float64),
                               wc.astype(np.float64), bc.astype(np.float64), orc.Storage.exact())
        e16 = max(e16, float(np.abs(o16[t] - exact).max()))
        ebf = max(ebf, float(np.abs(obf[t] - exact).max()))
    print(f"\n[level_fuse fp16 T{T} {H}x{W} level0={level0}] against the unrounded conv: fp16 form {e16:.2e}, bf16 form {ebf:.2e}")
    assert e16 < 0.25 * ebf


def test_level_fuse_fp16_maps_saturate(cuda):
    """A result beyond fp16's range is stored as +-65 504, not as inf (which every consumer would turn into NaN)."""
    import torch
    from slotvps_amd import ops
    T, H, W = 1, 16, 32
    cur = torch.full((T, 128, H, W), 3.0e3, device=cuda)
    cur[:, :, :, ::2] *= -1.0
    wc = torch.ones((256, 384), device=cuda).to(torch.float16)
    prev = torch.zeros((T, (H // 2) * (W // 2), 256), dtype=torch.float16, device=cuda)
    out = ops.level_fuse(cur, prev, wc, torch.zeros(256, device=cuda), H, W).float()
    assert torch.isfinite(out).all() and float(out.max()) == 65504.0 and float(out.min()) == -65504.0
    out0 = ops.level_fuse(cur, None, wc, torch.zeros(256, device=cuda), H, W).float()           # level-0 form (first kernel)
    assert torch.isfinite(out0).all() and float(out0.abs().max()) == 65504.0


@pytest.mark.parametrize("T,H,W,level0", [(2, 8, 16, True), (2, 16, 32, False), (2, 68, 120, False), (9, 32, 64, False), (3, 64, 128, False)])
@pytest.mark.parametrize("form", ["fp16", "bf16_in_fp16"])
def test_level_fuse_pixel_major_rows_in_the_fp16_forms(cuda, T, H, W, level0, form):
    """Round 4: the incoming map as 16-bit pixel-major rows (what the semantic tower's last kernel writes) in the fp16-encoded forms -
    fp16 rows for fp16 maps, bf16 rows for the bf16 policy in the fp16 encoding. The rows hold exactly the values the fp32 NCHW path
    would round its input to, so both paths must give the SAME bits (fast path, generic kernel and level 0)."""
    import torch
    from slotvps_amd import ops
    seed = 31 * H + W + T
    rng = np.random.default_rng(seed)
    params = synth.make_params({"conv_trans.conv.weight": (256, 384, 1, 1), "conv_trans.conv.bias": (256,)}, seed)
    twc = torch.from_numpy(params["conv_trans.conv.weight"].reshape(256, 384)).to(cuda)
    tbc = torch.from_numpy(params["conv_trans.conv.bias"]).to(cuda)
    cur = torch.from_numpy(np.stack([synth.smooth_features(rng, 128, H, W) for _ in range(T)])).to(cuda)
    rows_dt = torch.float16 if form == "fp16" else torch.bfloat16
    rows = cur.reshape(T, 128, H * W).transpose(1, 2).to(rows_dt).contiguous()                 # [T, HW, 128]
    nchw = rows.float().transpose(1, 2).reshape(T, 128, H, W).contiguous()                     # the same values as the reference's layout
    prev = None
    if not level0:
        prev = torch.from_numpy(np.stack([synth.smooth_features(rng, 256, H // 2, W // 2).reshape(256, -1).T * 1.5 for _ in range(T)])).to(cuda)
        prev = prev.to(torch.float16).contiguous() if form == "fp16" else prev.to(torch.bfloat16).float().to(torch.float16).contiguous()
    wc = twc.to(torch.float16).contiguous() if form == "fp16" else twc.to(torch.bfloat16).contiguous()
    a = ops.level_fuse(nchw, prev, wc, tbc, H, W, bf16_values=form == "bf16_in_fp16")
    b = ops.level_fuse(rows, prev, wc, tbc, H, W, bf16_values=form == "bf16_in_fp16")
    torch.cuda.synchronize()
    assert a.dtype == b.dtype == torch.float16 and torch.equal(a, b)
