"""Statistics-fused retriever (K3' retr_stats + K1' retr_attn + slot-side epilogue) against the float64 oracle.

Yardstick: `oracle.retriever` (the restatement of MaskDynamicConv.forward, pinned against the reference module) in
float64 with NO storage rounding, evaluated on the same bf16-stored feature map. The only differences left are the
kernel's own roundings: Q'' and P * rstd_v as bf16 hi + lo, the statistics operand f + pos and the QR factor as bf16,
fp32 accumulation. Tolerances are written at the assertions."""
import numpy as np
import pytest

from util import orc, to_bf16_t

pytestmark = pytest.mark.gpu


def make_module(cuda, seed):
    import torch
    from slotvps_amd.slot_head import MaskDynamicConv
    rng = np.random.default_rng(seed)
    m = MaskDynamicConv(256).to(cuda).eval()
    m.precision = "bf16"                                                         # this file: the 16-bit storage policy (the default is fp16x2)
    P = {}
    with torch.no_grad():
        for n in ("to_q", "to_k", "to_v"):
            lin = getattr(m, n)
            lim = float(np.sqrt(6.0 / 512))                                      # xavier-uniform
            P[f"{n}.weight"] = rng.uniform(-lim, lim, (256, 256)).astype(np.float32)
            P[f"{n}.bias"] = (0.1 * rng.standard_normal(256)).astype(np.float32)
            lin.weight.copy_(torch.from_numpy(P[f"{n}.weight"]))
            lin.bias.copy_(torch.from_numpy(P[f"{n}.bias"]))
        for n in ("norm_q", "norm_k", "norm_v", "norm1"):
            ln = getattr(m, n)
            P[f"{n}.weight"] = rng.uniform(0.5, 1.5, 256).astype(np.float32)
            P[f"{n}.bias"] = (0.1 * rng.standard_normal(256)).astype(np.float32)
            ln.weight.copy_(torch.from_numpy(P[f"{n}.weight"]))
            ln.bias.copy_(torch.from_numpy(P[f"{n}.bias"]))
    return m, P


@pytest.mark.parametrize("T,H,W,pos", [(2, 8, 32, True), (1, 5, 20, True), (1, 3, 64, False), (2, 34, 60, True), (1, 7, 9, True)])
def test_retr_stats(cuda, T, H, W, pos):
    import torch
    from slotvps_amd import ops
    m, P = make_module(cuda, H * W)
    rng = np.random.default_rng(W)
    HW = H * W
    feat = orc.round_bf16(rng.standard_normal((T, HW, 256)).astype(np.float32))
    c = m._fused_consts()
    tabs = ops.pos_embed_sine_tables(H, W, 256, cuda) if pos else None
    aux = ops.retr_stats(to_bf16_t(feat, cuda), H, W, m.retr_pos_tables(tabs), c["rk"], c["rbk"], 1e-5, c["rv"], c["rbv"], 1e-5)
    torch.cuda.synchronize()
    assert aux.shape == (T, HW, 8)                                               # ONE 16-byte row per pixel, nothing else is written
    rk, rv = (x.cpu().numpy() for x in ops.retr_stats_unpack(aux))
    aux_raw = aux.cpu().view(torch.int16).numpy().view(np.uint16)               # [T, HW, 8] fp16 bit patterns
    aux = aux.float().cpu().numpy()
    pm = orc.pos_embed_sine(H, W).astype(np.float64) if pos else 0.0
    d = lambda n: P[n].astype(np.float64)
    for t in range(T):
        f64 = feat[t].astype(np.float64)
        uk = orc.linear(f64 + pm, d("to_k.weight"), d("to_k.bias"))
        uv = orc.linear(f64, d("to_v.weight"), d("to_v.bias"))
        for got, u, bound in ((rk[t], uk, 1.5e-4), (rv[t], uv, 1e-3)):
            ref = 1.0 / np.sqrt(u.var(axis=1) + 1e-5)
            rel = np.abs(got - ref) / ref
            # key side fp16 x fp16 (2^-12 relative per element, averaged over ~100 random terms): measured ~2e-5 mean;
            # value side bf16 x bf16: measured 1e-4 mean / 4.7e-4 max (harmless there, see the kernel header)
            print(f"rstd rel err max {rel.max():.2e} mean {rel.mean():.2e} (bound {bound:.1e})")
            assert rel.max() <= bound, rel.max()
        sig = 1.0 / rv[t].astype(np.float64)
        assert np.all(aux[t][:, 0] == 1.0) and np.all(aux[t][:, 3] == 0.0)
        # bytes 8 .. 15 of the row: rstd_k, rstd_v as raw fp32 (what K1's producers read from the staged tile)
        packed = np.ascontiguousarray(aux_raw[t][:, 4:8]).view(np.float32)
        assert np.array_equal(packed[:, 0], rk[t]) and np.array_equal(packed[:, 1], rv[t])
        assert np.abs(aux[t][:, 1].astype(np.float64) + aux[t][:, 2] - sig).max() <= 3e-5 * sig.max()   # hi + lo: 16-bit mantissa


@pytest.mark.parametrize("T,H,W,L,pos", [(2, 8, 32, 100, True), (1, 16, 64, 128, True), (1, 5, 20, 37, True),
                                         (2, 34, 60, 100, True), (1, 3, 64, 1, False), (1, 40, 32, 100, True),
                                         (2, 34, 60, 200, True), (1, 16, 32, 256, True), (1, 9, 20, 129, False)])
def test_fused_retriever_matches_float64_oracle(cuda, T, H, W, L, pos):
    import torch
    from slotvps_amd import ops
    m, P = make_module(cuda, 7 + L)
    rng = np.random.default_rng(L + W)
    HW = H * W
    feat = orc.round_bf16(rng.standard_normal((T, HW, 256)).astype(np.float32))
    slots = rng.standard_normal((T, L, 256)).astype(np.float32)
    tabs = ops.pos_embed_sine_tables(H, W, 256, cuda) if pos else None
    with torch.no_grad():
        got = m.forward_fused(torch.from_numpy(slots).to(cuda), to_bf16_t(feat, cuda), (H, W), tabs)
        torch.cuda.synchronize()
        again = m.forward_fused(torch.from_numpy(slots).to(cuda), to_bf16_t(feat, cuda), (H, W), tabs)
    assert torch.equal(got, again)                                          # fixed-order partial sums: bitwise reproducible
    got = got.cpu().numpy()
    pm = orc.pos_embed_sine(H, W) if pos else None
    worst = 0.0
    for t in range(T):
        ref = orc.retriever(slots[t], feat[t], pm, P, "", st=orc.Storage.exact(), dt=np.float64)
        worst = max(worst, float(np.abs(got[t] - ref).max()))
    print(f"\nfused retriever T={T} {H}x{W} L={L}: max abs err vs float64 oracle {worst:.2e}")
    # post-LayerNorm outputs are O(1); measured 1e-4 ... 4e-4 typical, 1.6e-3 worst case (bf16 k / v tensors: 1e-1). The
    # largest contributions: rstd_k (fp16 x fp16 statistics, 2e-5 relative in front of logits of magnitude up to ~80) and
    # P * rstd_v carried as ONE fp16 on the value side (2^-12 relative per pixel, averaged over the pixel sum)
    assert worst <= 2e-3


@pytest.mark.parametrize("map_dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("name,T,H,W,L", [("R50 finest level, BASELINE config 1", 5, 256, 512, 100),
                                          ("VIPER finest level, BASELINE config 4", 10, 272, 480, 200),
                                          ("VIPER level 2 (width no multiple of the tile)", 10, 136, 240, 200)])
def test_full_size_sum_over_slots_properties(cuda, name, T, H, W, L, map_dtype):
    """Size-independent properties at the BASELINE sizes (the oracle does not finish there in seconds). The softmax runs
    over slots, so every pixel's column sums to one:
        sum_l A_l = sum_p rstd_v(p) f_p      sum_l s1_l = sum_p rstd_v(p)      sum_l s0_l = HW
    and the launch is bitwise reproducible."""
    import torch
    from slotvps_amd import ops
    m, _ = make_module(cuda, 3)
    g = torch.Generator(device=cuda).manual_seed(5)
    HW = H * W
    feat = torch.randn((T, HW, 256), generator=g, device=cuda).to(torch.float16 if map_dtype == "fp16" else torch.bfloat16)   # fp16 level maps: no conversion pass
    slots = torch.randn((T, L, 256), generator=g, device=cuda)
    tabs = ops.pos_embed_sine_tables(H, W, 256, cuda)
    c = m._fused_consts()
    with torch.no_grad():
        st = ops.retr_stats(feat, H, W, m.retr_pos_tables(tabs), c["rk"], c["rbk"], 1e-5, c["rv"], c["rbv"], 1e-5)
        q = ops.row_ln(m.to_q(slots), m.norm_q.weight, m.norm_q.bias, m.norm_q.eps)
        # the inputs of K1' exactly as MaskDynamicConv.forward_fused prepares them
        got = {}
        orig = ops.retr_attn
        def spy(*a, **k):
            got["ext"] = orig(*a, **k)
            return got["ext"]
        ops.retr_attn = spy
        try:
            out1 = m.forward_fused(slots, feat, (H, W), tabs, stats=st)
            ext1 = got["ext"].clone()
            out2 = m.forward_fused(slots, feat, (H, W), tabs, stats=st)
        finally:
            ops.retr_attn = orig
        torch.cuda.synchronize()
        assert torch.equal(out1, out2) and torch.isfinite(out1).all()
        tau = ops.retr_stats_unpack(st)[1].double()                             # rstd_v [T, HW]
        want_a = torch.einsum("tp,tpc->tc", tau, feat.double())                 # sum_p rstd_v f_p
        a_sum = ext1[:, :, :256].double().sum(1)
        scale = want_a.abs().max().item()
        # P * rstd_v is one fp16 per (slot, pixel): 2^-12 relative each, a random walk over 100 x 131 072 terms (measured 2.2e-4)
        assert (a_sum - want_a).abs().max().item() <= 5e-4 * max(scale, 1.0), ((a_sum - want_a).abs().max().item(), scale)
        s1 = ext1[:, :, 256].double().sum(1)
        s0 = ext1[:, :, 257].double().sum(1)
        assert ((s1 - tau.sum(1)).abs() / tau.sum(1)).max().item() <= 2e-5
        assert ((s0 - HW).abs() / HW).max().item() <= 2e-5


@pytest.mark.parametrize("T,H,W,pos,ns", [(2, 8, 32, True, 2), (1, 5, 20, True, 1), (1, 3, 64, False, 2), (2, 34, 60, True, 2),
                                          (1, 7, 9, True, 2), (3, 70, 96, True, 2), (8, 40, 64, True, 1)])
def test_level_statistics_equal_per_stage_statistics(cuda, T, H, W, pos, ns):
    """csrc/retr_stats2.hip (K3'': all stages of a level from one read of the map, eight waves sharing one LDS tile ring) against
    csrc/retr_stats.hip (one launch per stage): the same fp16 operands and fp32 tables; the partial sums of a wave cover the same 64
    rows in both kernels, so the two statistics agree to the order of the fp32 additions inside a wave (and bitwise in the fp16
    words derived from them almost everywhere)."""
    import torch
    from slotvps_amd import ops
    mods = [make_module(cuda, 100 * s + H * W)[0] for s in range(ns)]
    g = torch.Generator(device=cuda).manual_seed(W + ns)
    feat = torch.randn((T, H * W, 256), generator=g, device=cuda).to(torch.bfloat16)
    tabs = ops.pos_embed_sine_tables(H, W, 256, cuda) if pos else None
    with torch.no_grad():
        ref = [ops.retr_stats(feat, H, W, *m.stats_args(tabs)) for m in mods]
        got = ops.retr_stats_level(feat, H, W, [m.stats_args(tabs) for m in mods])
        again = ops.retr_stats_level(feat, H, W, [m.stats_args(tabs) for m in mods])
    torch.cuda.synchronize()
    for s in range(ns):
        assert torch.equal(got[s].view(torch.int16), again[s].view(torch.int16))      # (bit patterns: the fp32 words are not fp16 numbers)
        rk0, rv0 = (x.double() for x in ops.retr_stats_unpack(ref[s]))
        rk1, rv1 = (x.double() for x in ops.retr_stats_unpack(got[s]))
        ek = ((rk1 - rk0).abs() / rk0).max().item()
        ev = ((rv1 - rv0).abs() / rv0).max().item()
        print(f"\nstage {s}: rstd_k rel diff {ek:.2e}, rstd_v rel diff {ev:.2e}")
        assert ek <= 2e-6 and ev <= 2e-6                       # fp32 summation order of 64 squares only
        a0, a1 = ref[s].float(), got[s].float()
        assert torch.equal(a0[..., 0], a1[..., 0]) and torch.equal(a0[..., 3], a1[..., 3])          # the constant words 1, 0
        sig0 = a0[..., 1].double() + a0[..., 2].double()
        sig1 = a1[..., 1].double() + a1[..., 2].double()
        assert ((sig1 - sig0).abs() / sig0).max().item() <= 4e-5                                    # hi + lo of sigma_v: 16-bit mantissa


def test_statistics_against_float64_layernorm(cuda):
    """rstd_k / rstd_v of K3' (fp16 QR factors) against a float64 evaluation of the reference's LayerNorm statistics
    (dynamic_mask_head.py:432-433): <= 1e-4 relative (measured ~3e-5 / ~7e-5); the fused retriever against the float64 oracle on the
    same bf16 map: <= 2e-3 (measured 1.0e-3 ... 1.7e-3). The reference-precision forms of both (hi + lo factors, hi + lo probabilities on
    hi + lo planes) are held to 1e-6 / 2e-4 in tests/test_refprec_gpu.py."""
    import torch
    from slotvps_amd import ops
    from slotvps_amd.slot_head import MaskDynamicConv
    rng = np.random.default_rng(7)
    m = MaskDynamicConv(256).to(cuda).eval()
    m.precision = "bf16"                                                         # this file: the 16-bit storage policy (the default is fp16x2)
    P = {}
    with torch.no_grad():
        for n in ("to_q", "to_k", "to_v"):
            lim = float(np.sqrt(6.0 / 512))
            P[f"{n}.weight"] = rng.uniform(-lim, lim, (256, 256)).astype(np.float32)
            P[f"{n}.bias"] = (0.1 * rng.standard_normal(256)).astype(np.float32)
            getattr(m, n).weight.copy_(torch.from_numpy(P[f"{n}.weight"]))
            getattr(m, n).bias.copy_(torch.from_numpy(P[f"{n}.bias"]))
        for n in ("norm_q", "norm_k", "norm_v", "norm1"):
            P[f"{n}.weight"] = rng.uniform(0.5, 1.5, 256).astype(np.float32)
            P[f"{n}.bias"] = (0.1 * rng.standard_normal(256)).astype(np.float32)
            getattr(m, n).weight.copy_(torch.from_numpy(P[f"{n}.weight"]))
            getattr(m, n).bias.copy_(torch.from_numpy(P[f"{n}.bias"]))
    worst = {}
    for (H, W, L) in ((12, 40, 100), (32, 64, 100), (9, 33, 37), (16, 64, 128)):
        feat = orc.round_bf16(rng.standard_normal((2, H * W, 256)).astype(np.float32))
        slots = rng.standard_normal((2, L, 256)).astype(np.float32)
        tabs = ops.pos_embed_sine_tables(H, W, 256, cuda)
        ft = torch.from_numpy(feat).to(cuda).to(torch.bfloat16)
        pos = orc.pos_embed_sine(H, W).astype(np.float64)                                      # [HW, 256]
        with torch.no_grad():
            aux_f = ops.retr_stats(ft, H, W, *m.stats_args(tabs))
            torch.cuda.synchronize()
            rk_f, rv_f = (x.cpu().numpy().astype(np.float64) for x in ops.retr_stats_unpack(aux_f))
            a16 = aux_f.cpu().numpy()
            assert (a16[..., 0] == 1.0).all() and (a16[..., 3] == 0.0).all()                 # the constant words of the aux rows
        for t in range(2):
            x = feat[t].astype(np.float64)
            k = (x + pos) @ P["to_k.weight"].astype(np.float64).T + P["to_k.bias"].astype(np.float64)
            v = x @ P["to_v.weight"].astype(np.float64).T + P["to_v.bias"].astype(np.float64)
            ref_k = 1.0 / np.sqrt(k.var(axis=1) + 1e-5)
            ref_v = 1.0 / np.sqrt(v.var(axis=1) + 1e-5)
            for name, got, ref in (("rstd_k", rk_f[t], ref_k), ("rstd_v", rv_f[t], ref_v)):
                worst[name] = max(worst.get(name, 0), float(np.abs(got / ref - 1).max()))
        with torch.no_grad():
            got = m.forward_fused(torch.from_numpy(slots).to(cuda), ft, (H, W), tabs).cpu().numpy()
            for t in range(2):
                ref = orc.retriever(slots[t], feat[t], orc.pos_embed_sine(H, W), P, "", st=orc.Storage.exact(), dt=np.float64)
                worst["retriever"] = max(worst.get("retriever", 0), float(np.abs(got[t] - ref).max()))
    print("\n[statistics / fused retriever against float64] " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))
    assert worst["rstd_k"] <= 1e-4 and worst["rstd_v"] <= 2e-4, worst
    assert worst["retriever"] <= 2e-3, worst


@pytest.mark.parametrize("H,W,L", [(12, 40, 100), (32, 64, 100), (9, 33, 37), (16, 64, 200)])
def test_fp16_map_equals_bf16_map_on_bf16_values(cuda, H, W, L):
    """An fp16 level map (MultiScaleDynamicMaskHead.map_dtype = "fp16") makes the statistics / retriever kernels skip their bf16 -> fp16
    pass in LDS. With map values that are exactly representable in both formats the LDS tiles are the same bits: the aux rows of K3',
    K3'' and the retriever's result must be bit-identical to the bf16-map run."""
    import torch
    from slotvps_amd import ops
    from slotvps_amd.slot_head import MaskDynamicConv
    rng = np.random.default_rng(H * W + L)
    torch.manual_seed(5)
    m = MaskDynamicConv(256).to(cuda).eval()
    m.precision = "bf16"                                                         # this file: the 16-bit storage policy (the default is fp16x2)
    m2 = MaskDynamicConv(256).to(cuda).eval()
    m2.precision = "bf16"
    feat = orc.round_bf16(rng.standard_normal((2, H * W, 256)).astype(np.float32))
    feat[np.abs(feat) < 2.0 ** -13] = 0.0                  # below fp16's normal range a bf16 value is not an fp16 value (or a subnormal one)
    slots = torch.from_numpy(rng.standard_normal((2, L, 256)).astype(np.float32)).to(cuda)
    tabs = ops.pos_embed_sine_tables(H, W, 256, cuda)
    fb = torch.from_numpy(feat).to(cuda).to(torch.bfloat16)
    fh = torch.from_numpy(feat).to(cuda).to(torch.float16)
    assert torch.equal(fb.float(), fh.float())
    with torch.no_grad():
        c = m._fused_consts()
        args = m.stats_args(tabs)
        bits = lambda t: t.view(torch.int16)                   # the aux rows carry raw fp32 words: compare bits, not fp16 values (NaN patterns)
        assert torch.equal(bits(ops.retr_stats(fb, H, W, *args)), bits(ops.retr_stats(fh, H, W, *args)))
        pair_b = ops.retr_stats_level(fb, H, W, [m.stats_args(tabs), m2.stats_args(tabs)])
        pair_h = ops.retr_stats_level(fh, H, W, [m.stats_args(tabs), m2.stats_args(tabs)])
        assert all(torch.equal(bits(a), bits(b)) for a, b in zip(pair_b, pair_h))
        assert torch.equal(m.forward_fused(slots, fb, (H, W), tabs), m.forward_fused(slots, fh, (H, W), tabs))
