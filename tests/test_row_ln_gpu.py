"""K5 parity: fused residual + LayerNorm (+ReLU) (+residual) (+bf16 cast) rows vs the oracle's layer_norm."""
import numpy as np
import pytest

from util import orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,groups,relu,pre,post,bf16", [
    (500, 1, False, True, False, False),
    (500, 1, True, False, False, False),
    (1500, 3, False, False, False, False),      # grouped affines (q / k / v of the temporal retriever)
    (1000, 2, True, False, False, False),       # class + embedding tower layer
    (500, 1, False, True, True, False),         # S + LN3'(U + y)
    (100, 1, False, False, False, True),        # q -> bf16
    (7, 1, True, True, True, False),            # rows not a multiple of the 4 rows per workgroup
])
def test_row_ln_matches_oracle(cuda, rows, groups, relu, pre, post, bf16):
    import torch
    from slotvps_amd import ops
    rng = np.random.default_rng(rows + groups)
    x = (3.0 * rng.standard_normal((rows, 256)) + 0.5).astype(np.float32)
    p = rng.standard_normal((rows, 256)).astype(np.float32)
    q = rng.standard_normal((rows, 256)).astype(np.float32)
    w = rng.uniform(0.5, 1.5, (groups, 256)).astype(np.float32)
    b = (0.1 * rng.standard_normal((groups, 256))).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(cuda)
    got = ops.row_ln(t(x), t(w), t(b), 1e-5, pre=t(p) if pre else None, post=t(q) if post else None, relu=relu,
                     rows_per_group=-(-rows // groups), out_bf16=bf16)
    torch.cuda.synchronize()
    got = got.float().cpu().numpy()
    rpg = -(-rows // groups)
    gidx = np.arange(rows) // rpg
    xin = (x + p if pre else x).astype(np.float64)
    ref = orc.layer_norm(xin, w[gidx].astype(np.float64), b[gidx].astype(np.float64))
    if relu:
        ref = np.maximum(ref, 0)
    if post:
        ref = ref + q
    tol = 2e-2 if bf16 else 2e-5          # bf16 output: one ulp at |y| < 4; fp32: accumulation order only
    assert np.abs(got - ref).max() < tol


@pytest.mark.parametrize("T,L", [(3, 100), (2, 37), (1, 200), (1, 256)])
def test_slot_self_attention_kernel_matches_oracle_mha(cuda, T, L):
    """svps_slot_self_attn + the two projections == the oracle's nn.MultiheadAttention restatement (float64)."""
    import torch
    from slotvps_amd import ops
    from util import orc
    rng = np.random.default_rng(L)
    P = {"in_proj_weight": (rng.standard_normal((768, 256)) / 16).astype(np.float32),
         "in_proj_bias": (0.1 * rng.standard_normal(768)).astype(np.float32),
         "out_proj.weight": (rng.standard_normal((256, 256)) / 16).astype(np.float32),
         "out_proj.bias": (0.1 * rng.standard_normal(256)).astype(np.float32)}
    x = rng.standard_normal((T, L, 256)).astype(np.float32)
    tx = torch.from_numpy(x).to(cuda)
    g = lambda n: torch.from_numpy(P[n]).to(cuda)
    qkv = torch.nn.functional.linear(tx, g("in_proj_weight"), g("in_proj_bias"))
    o = ops.slot_self_attn(qkv.contiguous(), 8)
    got = torch.nn.functional.linear(o, g("out_proj.weight"), g("out_proj.bias")).cpu().numpy()
    o16 = ops.slot_self_attn(qkv.contiguous(), 8, split="fp16")       # fp16 hi + lo operands: the slot side of mode fp16x2
    got16 = torch.nn.functional.linear(o16.double(), g("out_proj.weight").double(), g("out_proj.bias").double()).cpu().numpy()
    qkv64 = (x.astype(np.float64) @ P["in_proj_weight"].astype(np.float64).T + P["in_proj_bias"].astype(np.float64))
    for t in range(T):
        ref = orc.multihead_self_attention(x[t].astype(np.float64), P, "", 8, np.float64)
        assert np.abs(got[t] - ref).max() <= 5e-5          # split-bf16 products with fp32 accumulation: fp32-class
        # the kernel alone in the fp16 split against float64 on the SAME fp32 qkv (the projections in float64): the operand split's 22 bits
        q_, k_, v_ = (qkv[t].double().cpu().numpy()[:, i * 256:(i + 1) * 256].reshape(L, 8, 32).transpose(1, 0, 2) for i in range(3))
        a_ = q_ @ k_.transpose(0, 2, 1) / np.sqrt(32.0)
        a_ = np.exp(a_ - a_.max(-1, keepdims=True))
        o_ = ((a_ / a_.sum(-1, keepdims=True)) @ v_).transpose(1, 0, 2).reshape(L, 256)
        e16 = float(np.abs(o16[t].double().cpu().numpy() - o_).max())
        assert e16 <= 3e-6, e16                              # (the bf16 split on the same inputs: ~2e-5)
        assert np.abs(got16[t] - ref).max() <= 2e-5
    del qkv64


def test_retr_query_prep_and_split(cuda):
    import torch
    from slotvps_amd import ops
    from util import orc
    rng = np.random.default_rng(1)
    T, L, LP = 2, 100, 128
    x = rng.standard_normal((T, L, 256)).astype(np.float32)
    v = {n: rng.uniform(0.5, 1.5, 256).astype(np.float32) if n.endswith("w") else (0.1 * rng.standard_normal(256)).astype(np.float32)
         for n in ("qw", "qb", "kw", "kb", "bck")}
    g = lambda n: torch.from_numpy(v[n]).to(cuda)
    gp, c3, a1 = ops.retr_query_prep(torch.from_numpy(x).to(cuda), g("qw"), g("qb"), 1e-5, g("kw"), g("kb"), g("bck"), LP)
    q = orc.layer_norm(x.astype(np.float64), v["qw"].astype(np.float64), v["qb"].astype(np.float64))
    gr = q * v["kw"]
    assert gp.shape == (T, LP, 256) and torch.all(gp[:, L:] == 0) and torch.all(c3[:, L:] <= -1e30) and torch.all(a1[:, L:] == 0)
    assert np.abs(gp[:, :L].cpu().numpy() - gr).max() <= 1e-5
    assert np.abs(c3[:, :L].cpu().numpy() - np.log2(np.e) * (q @ v["kb"].astype(np.float64))).max() <= 4e-5
    assert np.abs(a1[:, :L].cpu().numpy() - gr @ v["bck"].astype(np.float64)).max() <= 2e-5
    hi, lo = ops.retr_split(gp)
    assert torch.equal(hi, gp.to(torch.float16)) and torch.equal(lo, (gp - hi.float()).to(torch.float16))
    assert (hi.float() + lo.float() - gp).abs().max().item() <= 2.0 ** -21 * gp.abs().max().item()


@pytest.mark.parametrize("M,K,N,act,bias", [(8000, 256, 768, None, True), (500, 2048, 256, None, True), (1000, 256, 2048, "gelu", True),
                                            (37, 272, 256, None, False), (200, 256, 1024, "relu", True), (65, 256, 256, None, False)])
def test_slot_gemm_matches_float64_linear(cuda, M, K, N, act, bias):
    """K8 (csrc/slot_gemm.hip): y = act(x W^T + b) with split-bf16 matrix-core products against float64: fp32-class."""
    import torch
    from slotvps_amd import ops
    from util import orc
    rng = np.random.default_rng(M + K + N)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = (0.1 * rng.standard_normal(N)).astype(np.float32) if bias else None
    wp = ops.pack_b_fragments(torch.from_numpy(w).to(cuda))
    code = {None: ops.ACT_NONE, "relu": ops.ACT_RELU, "gelu": ops.ACT_GELU}[act]
    got = ops.slot_gemm(torch.from_numpy(x).to(cuda), wp, None if b is None else torch.from_numpy(b).to(cuda), code)
    again = ops.slot_gemm(torch.from_numpy(x).to(cuda), wp, None if b is None else torch.from_numpy(b).to(cuda), code)
    assert torch.equal(got, again)
    ref = x.astype(np.float64) @ w.astype(np.float64).T + (0 if b is None else b.astype(np.float64))
    if act == "relu":
        ref = np.maximum(ref, 0)
    elif act == "gelu":
        ref = orc.gelu(ref)
    err = np.abs(got.cpu().numpy() - ref).max()
    print(f"\nK8 M={M} K={K} N={N} act={act}: max abs err vs float64 {err:.2e} (outputs of order 1)")
    assert err <= 3e-5


@pytest.mark.parametrize("M,K,N", [(16000, 256, 256), (300, 256, 256), (37, 272, 512)])
def test_slot_gemm_fp16_split_is_fp32_class(cuda, M, K, N):
    """K8 with both operands split into fp16 hi + lo (svps_slot_gemm_f16: the query side of the fused retriever): 22 bits of
    mantissa for the same three MFMAs - against float64 an order of magnitude closer than the bf16 split, and closer than an fp32
    GEMM accumulating in fp32."""
    import torch
    from slotvps_amd import ops
    rng = np.random.default_rng(M + K + N)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = (0.1 * rng.standard_normal(N)).astype(np.float32)
    tx, tb = torch.from_numpy(x).to(cuda), torch.from_numpy(b).to(cuda)
    ref = x.astype(np.float64) @ w.astype(np.float64).T + b.astype(np.float64)
    got16 = ops.slot_gemm(tx, ops.pack_b_fragments(torch.from_numpy(w).to(cuda), split="fp16"), tb)
    gotbf = ops.slot_gemm(tx, ops.pack_b_fragments(torch.from_numpy(w).to(cuda)), tb)
    assert torch.equal(got16, ops.slot_gemm(tx, ops.pack_b_fragments(torch.from_numpy(w).to(cuda), split="fp16"), tb))
    e16, ebf = np.abs(got16.cpu().numpy() - ref).max(), np.abs(gotbf.cpu().numpy() - ref).max()
    print(f"\nK8 M={M} K={K} N={N}: fp16 split {e16:.2e}, bf16 split {ebf:.2e} (outputs of order 1)")
    assert e16 <= 5e-6 and e16 < 0.3 * ebf                       # measured 2.8e-6 against 2.5e-5: what is left is the fp32 accumulation
    # round 4: the fp16-split form has the epilogues of the bf16-split form - activation, LayerNorm, the one-launch FFN (precision "fp16x2")
    wp16 = ops.pack_b_fragments(torch.from_numpy(w).to(cuda), split="fp16")
    assert torch.equal(ops.slot_gemm(tx, wp16, tb, ops.ACT_RELU), torch.relu(got16))
    gl = ops.slot_gemm(tx, wp16, tb, ops.ACT_GELU).cpu().numpy()
    from scipy.special import erf
    assert np.abs(gl - 0.5 * ref * (1.0 + erf(ref / np.sqrt(2.0)))).max() <= 8e-6
    if N == 256:
        gamma = torch.from_numpy(rng.uniform(0.5, 1.5, 256).astype(np.float32)).to(cuda)
        beta = torch.from_numpy((0.1 * rng.standard_normal(256)).astype(np.float32)).to(cuda)
        pre = torch.from_numpy(rng.standard_normal((M, 256)).astype(np.float32)).to(cuda)
        fused = ops.slot_gemm_ln(tx, wp16, tb, gamma, beta, 1e-5, pre=pre, relu=True)
        two = ops.row_ln(got16, gamma, beta, 1e-5, pre=pre, relu=True)
        assert torch.equal(fused, two)                                # bitwise the two-launch form, as for the bf16 split
    with pytest.raises(ValueError):
        ops.pack_b_fragments(torch.full((256, 256), 7e4, device=cuda), split="fp16")


@pytest.mark.parametrize("M,H,act", [(16000, 2048, "gelu"), (500, 1024, "relu"), (70, 512, "gelu")])
def test_slot_ffn_fp16_split_is_bitwise_the_two_launches(cuda, M, H, act):
    """The one-launch feed-forward block in the fp16-split form (svps_slot_ffn_f16, precision "fp16x2") == svps_slot_gemm_f16_act
    followed by svps_slot_gemm_ln_f16 (bit for bit with ReLU; with GELU a few fp32 ulps on ~1.5 % of the outputs, measured 3.6e-7: the
    two kernels do not round every fp16 lo part of the small GELU values alike); and fp32-class against float64."""
    import torch
    from scipy.special import erf
    from slotvps_amd import ops
    rng = np.random.default_rng(M + H)
    x = rng.standard_normal((M, 256)).astype(np.float32)
    w1 = (rng.standard_normal((H, 256)) / 16).astype(np.float32)
    w2 = (rng.standard_normal((256, H)) / np.sqrt(H)).astype(np.float32)
    b1, b2 = (0.1 * rng.standard_normal(H)).astype(np.float32), (0.1 * rng.standard_normal(256)).astype(np.float32)
    gamma, beta = rng.uniform(0.5, 1.5, 256).astype(np.float32), (0.1 * rng.standard_normal(256)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(cuda)
    tx = t(x)
    p1, p2 = ops.pack_b_fragments(t(w1), split="fp16"), ops.pack_b_fragments(t(w2), split="fp16")
    code = ops.ACT_GELU if act == "gelu" else ops.ACT_RELU
    one = ops.slot_ffn(tx, p1, t(b1), p2, t(b2), t(gamma), t(beta), 1e-5, act=code, pre=tx)
    hid = ops.slot_gemm(tx, p1, t(b1), code)
    two = ops.slot_gemm_ln(hid, p2, t(b2), t(gamma), t(beta), 1e-5, pre=tx)
    assert torch.equal(one, two) if act == "relu" else (one - two).abs().max().item() <= 1.5e-6
    h64 = x.astype(np.float64) @ w1.astype(np.float64).T + b1
    h64 = 0.5 * h64 * (1.0 + erf(h64 / np.sqrt(2.0))) if act == "gelu" else np.maximum(h64, 0.0)
    y = x + h64 @ w2.astype(np.float64).T + b2
    y = (y - y.mean(-1, keepdims=True)) / np.sqrt(y.var(-1, keepdims=True) + 1e-5) * gamma + beta
    err = np.abs(one.cpu().numpy() - y).max()
    print(f"\nFFN fp16 split M={M} H={H} {act}: {err:.2e} against float64")
    assert err <= 1.5e-5


@pytest.mark.parametrize("M,K,pre,post,relu,bias", [(16000, 256, True, False, False, True), (500, 2048, True, True, False, True),
                                                    (37, 272, False, False, True, False), (8000, 256, False, False, True, False),
                                                    (65, 256, True, False, True, True), (40000, 256, True, False, False, True)])
def test_slot_gemm_ln_is_bitwise_gemm_then_row_ln(cuda, M, K, pre, post, relu, bias):
    """K8 with the LayerNorm epilogue (svps_slot_gemm_ln) == svps_slot_gemm followed by svps_row_ln, bit for bit (same products,
    same LayerNorm arithmetic), for the three launch shapes (32-row tiles, 64-row tiles, the wide variant) and the long-K path;
    and against a float64 LayerNorm of the float64 product."""
    import torch
    from slotvps_amd import ops
    g = torch.Generator(device=cuda).manual_seed(M + K)
    x = torch.randn((M, K), generator=g, device=cuda)
    w = torch.randn((256, K), generator=g, device=cuda) / K ** 0.5
    b = 0.1 * torch.randn((256,), generator=g, device=cuda) if bias else None
    gamma = 1.0 + 0.1 * torch.randn((256,), generator=g, device=cuda)
    beta = 0.1 * torch.randn((256,), generator=g, device=cuda)
    p1 = torch.randn((M, 256), generator=g, device=cuda) if pre else None
    p2 = torch.randn((M, 256), generator=g, device=cuda) if post else None
    wp = ops.pack_b_fragments(w)
    two = ops.row_ln(ops.slot_gemm(x, wp, b), gamma, beta, 1e-5, pre=p1, post=p2, relu=relu)
    one = ops.slot_gemm_ln(x, wp, b, gamma, beta, 1e-5, pre=p1, post=p2, relu=relu)
    assert torch.equal(one, two)
    y = x.double() @ w.double().t() + (0 if b is None else b.double())
    y = y + (0 if p1 is None else p1.double())
    y = (y - y.mean(1, keepdim=True)) / torch.sqrt(y.var(1, unbiased=False, keepdim=True) + 1e-5) * gamma.double() + beta.double()
    y = torch.relu(y) if relu else y
    y = y + (0 if p2 is None else p2.double())
    assert (one.double() - y).abs().max().item() <= 1e-4


def test_row_softmax_matches_torch(cuda):
    """The temporal retriever's softmax over the query axis (dynamic_mask_head.py:559-567) as the library's own kernel."""
    import torch
    from slotvps_amd import ops
    g = torch.Generator(device=cuda).manual_seed(3)
    for shape in ((32, 500, 500), (3, 7, 1), (5, 64), (2, 3, 1000)):
        x = 20.0 * torch.randn(shape, generator=g, device=cuda)
        ref = torch.softmax(x.double(), dim=-1)
        y = ops.row_softmax(x)
        assert float((y.double() - ref).abs().max()) < 5e-7
        assert float((y.sum(-1) - 1).abs().max()) < 1e-5
        z = x.clone()
        assert ops.row_softmax(z, inplace=True) is z and torch.equal(z, y)


@pytest.mark.parametrize("M,H,act,post", [
    (1000, 2048, "gelu", False),     # MaskRCNNHead's block (dynamic_mask_head.py:379-385); 1000 rows: a ragged last 64-row tile
    (1000, 1024, "relu", True),      # TemporalSlotsHead's block with the caller's residual (:519-525, :317)
    (64, 256, "relu", False),        # one hidden chunk, one tile
    (3, 512, "gelu", True),
])
def test_slot_ffn_is_bitwise_the_two_launch_form(cuda, M, H, act, post):
    """The one-launch feed-forward block (csrc/slot_ffn.hip) against svps_slot_gemm (activation) + svps_slot_gemm_ln on the same
    packed weights - bit for bit - and against a float64 evaluation of LN(x + W2 act(W1 x + b1) + b2)."""
    import torch
    import torch.nn.functional as F
    from slotvps_amd import ops
    g = torch.Generator(device=cuda).manual_seed(M + H)
    x = torch.randn((M, 256), generator=g, device=cuda)
    w1 = torch.randn((H, 256), generator=g, device=cuda) / 16.0
    b1 = 0.1 * torch.randn((H,), generator=g, device=cuda)
    w2 = torch.randn((256, H), generator=g, device=cuda) / H ** 0.5
    b2 = 0.1 * torch.randn((256,), generator=g, device=cuda)
    gamma = torch.rand((256,), generator=g, device=cuda) + 0.5
    beta = 0.1 * torch.randn((256,), generator=g, device=cuda)
    q = torch.randn((M, 256), generator=g, device=cuda) if post else None
    p1, p2 = ops.pack_b_fragments(w1), ops.pack_b_fragments(w2)
    code = ops.ACT_RELU if act == "relu" else ops.ACT_GELU
    hid = ops.slot_gemm(x, p1, b1, code)
    two = ops.slot_gemm_ln(hid, p2, b2, gamma, beta, 1e-5, pre=x, post=q)
    one = ops.slot_ffn(x, p1, b1, p2, b2, gamma, beta, 1e-5, act=code, pre=x, post=q)
    torch.cuda.synchronize()
    assert torch.equal(one, two), f"fused FFN differs from the two-launch form: {float((one - two).abs().max())}"
    f = F.relu if act == "relu" else F.gelu
    xd = x.double()
    ref = F.layer_norm(xd + F.linear(f(F.linear(xd, w1.double(), b1.double())), w2.double(), b2.double()), (256,),
                       gamma.double(), beta.double(), 1e-5)
    if post:
        ref = ref + q.double()
    assert float((one.double() - ref).abs().max()) < 1e-4


@pytest.mark.parametrize("M", [1000, 64, 5])
def test_slot_chain_is_bitwise_the_per_layer_launches(cuda, M):
    """csrc/slot_chain.hip on the three chain shapes of the slot update - a tower (cls layer and three embedding layers off one
    input, dynamic_mask_head.py:394-397), three projections of one input (:555-557), out_proj + residual + norm followed by a
    projection + norm (:356-358, :431) - against svps_slot_gemm_ln launched per layer on the same packed weights, bit for bit."""
    import torch
    from slotvps_amd import ops
    g = torch.Generator(device=cuda).manual_seed(M)
    rnd = lambda *s: torch.randn(s, generator=g, device=cuda)
    x, res, post = rnd(M, 256), rnd(M, 256), rnd(M, 256)
    W = [ops.pack_b_fragments(rnd(256, 256) / 16.0) for _ in range(4)]
    B = [0.1 * rnd(256) for _ in range(4)]
    G = [torch.rand(256, generator=g, device=cuda) + 0.5 for _ in range(4)]
    E = [0.1 * rnd(256) for _ in range(4)]
    # tower: reg0 -> reg1 -> reg2 (stored), cls0 off the input (stored)
    r0 = ops.slot_gemm_ln(x, W[0], None, G[0], E[0], 1e-5, relu=True)
    r1 = ops.slot_gemm_ln(r0, W[1], None, G[1], E[1], 1e-5, relu=True)
    r2 = ops.slot_gemm_ln(r1, W[2], None, G[2], E[2], 1e-5, relu=True)
    c0 = ops.slot_gemm_ln(x, W[3], None, G[3], E[3], 1e-5, relu=True)
    outs = ops.slot_chain(x, [dict(wpack=W[0], gamma=G[0], beta=E[0], relu=True),
                              dict(wpack=W[1], gamma=G[1], beta=E[1], relu=True),
                              dict(wpack=W[2], gamma=G[2], beta=E[2], relu=True, out=True),
                              dict(wpack=W[3], gamma=G[3], beta=E[3], relu=True, src="x")])
    torch.cuda.synchronize()
    assert outs[0] is None and outs[1] is None
    assert torch.equal(outs[2], r2) and torch.equal(outs[3], c0)
    # three projections of one input, with bias
    ref = [ops.slot_gemm_ln(x, W[i], B[i], G[i], E[i], 1e-5) for i in range(3)]
    qkv = torch.empty((3, M, 256), device=cuda)
    ops.slot_chain(x, [dict(wpack=W[i], bias=B[i], gamma=G[i], beta=E[i], src="x", out=qkv[i]) for i in range(3)])
    torch.cuda.synchronize()
    assert all(torch.equal(qkv[i], ref[i]) for i in range(3))
    # projection + residual + norm (stored), then projection + norm + post of that result
    a1 = ops.slot_gemm_ln(x, W[0], B[0], G[0], E[0], 1e-5, pre=res)
    a2 = ops.slot_gemm_ln(a1, W[1], B[1], G[1], E[1], 1e-6, post=post)
    o = ops.slot_chain(x, [dict(wpack=W[0], bias=B[0], gamma=G[0], beta=E[0], pre=res, out=True),
                           dict(wpack=W[1], bias=B[1], gamma=G[1], beta=E[1], eps=1e-6, post=post)])
    torch.cuda.synchronize()
    assert torch.equal(o[0], a1) and torch.equal(o[1], a2)
