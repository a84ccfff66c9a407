"""K9 (csrc/bgemm.hip): C[g, m, n] = alpha * sum_k A[g, m, k] B[g, n, k] (+ bias[g, n]) on arbitrary strides against float64.
The shapes are the ones the slot side uses it for (temporal slot retriever, dynamic_mask_head.py:550-572; position terms of the
fused retriever; class projection :398) plus ragged / transposed / broadcast cases."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _check(got, a, b, bias, alpha, bound=2e-5):
    import torch
    want = alpha * torch.matmul(a.double(), b.double().transpose(-1, -2))
    if bias is not None:
        want = want + (bias.double().unsqueeze(-2) if bias.dim() >= 1 else bias.double())
    err = (got.double() - want).abs().max().item()
    scale = max(want.abs().max().item(), 1e-30)
    # both operands carried as bf16 hi + lo, three products: ~2^-16 relative per term, fp32 accumulation
    assert err <= bound * scale, (err, scale)
    return err / scale


@pytest.mark.parametrize("G,M,N,K", [(3, 70, 50, 40), (1, 1, 1, 1), (2, 64, 64, 32), (2, 500, 500, 256), (2, 33, 129, 17),
                                     (1, 1600, 20, 256), (5, 16, 128, 128),
                                     (2, 2000, 2000, 256), (1, 2000, 256, 2000)])     # VIPER T = 10 x L = 200: logits [TL, TL] and attn^T v (K = TL)
def test_bgemm_contiguous(cuda, G, M, N, K):
    import torch
    from slotvps_amd import ops
    g = torch.Generator(device=cuda).manual_seed(G * 1000 + M)
    a = torch.randn((G, M, K), generator=g, device=cuda)
    b = torch.randn((G, N, K), generator=g, device=cuda)
    bias = torch.randn((G, N), generator=g, device=cuda)
    print("rel err", _check(ops.bgemm(a, b), a, b, None, 1.0), _check(ops.bgemm(a, b, bias=bias, alpha=0.5), a, b, bias, 0.5))


def test_bgemm_fp16_split(cuda):
    """svps_bgemm_f16: the position terms of the fused retriever (sine tables x folded queries) with fp16 hi + lo operands."""
    import torch
    from slotvps_amd import ops
    g = torch.Generator(device=cuda).manual_seed(7)
    tab = torch.sin(torch.randn((64, 128), generator=g, device=cuda))             # shared table: batch stride 0
    q2 = 3.0 * torch.randn((6, 128, 256), generator=g, device=cuda)
    a1 = torch.randn((6, 128), generator=g, device=cuda)
    for split, bound in (("fp16", 3e-7), ("bf16", 4e-5)):
        got = ops.bgemm(tab, q2[:, :, :128], bias=a1, split=split)
        want = torch.matmul(tab.double(), q2[:, :, :128].double().transpose(1, 2)) + a1.double()[:, None, :]
        err = (got.double() - want).abs().max().item() / want.abs().max().item()
        print(f"bgemm split {split}: rel err {err:.2e}")
        assert err <= bound, (split, err)


def test_bgemm_strides_and_broadcast(cuda):
    import torch
    from slotvps_amd import ops
    g = torch.Generator(device=cuda).manual_seed(7)
    G, M, N, K = 4, 100, 96, 72
    # A stored k-major (a transposed view), B stored k-major, C written into a strided slice of a larger buffer
    at = torch.randn((G, K, M), generator=g, device=cuda)
    bt = torch.randn((G, K, N), generator=g, device=cuda)
    a, b = at.transpose(1, 2), bt.transpose(1, 2)
    big = torch.zeros((G, M, 2 * N + 3), device=cuda)
    out = big[:, :, 3:3 + 2 * N:2]
    ops.bgemm(a, b, out=out)
    _check(out, a, b, None, 1.0)
    assert big[:, :, 0:3].abs().max().item() == 0.0 and big[:, :, 4::2].abs().max().item() == 0.0      # nothing else touched
    # shared A (2-D), per-batch B as a column slice of a wider tensor, bias [G, N] as a column slice, bias [N]
    a2 = torch.randn((M, K), generator=g, device=cuda)
    wide = torch.randn((G, N, 2 * K), generator=g, device=cuda)
    bsl = wide[:, :, K:]
    biasw = torch.randn((G, 2 * N), generator=g, device=cuda)
    _check(ops.bgemm(a2, bsl, bias=biasw[:, :N]), a2.unsqueeze(0).expand(G, -1, -1), bsl, biasw[:, :N], 1.0)
    bias1 = torch.randn((N,), generator=g, device=cuda)
    _check(ops.bgemm(a2.unsqueeze(0).expand(G, -1, -1), bsl, bias=bias1), a2.unsqueeze(0).expand(G, -1, -1), bsl,
           bias1.unsqueeze(0).expand(G, -1), 1.0)


def test_bgemm_temporal_retriever_shapes(cuda):
    """softmax(k q^T) over the last axis, then out[lq, c] = sum_lk attn_t[lk, lq] v[lk, c] (A read k-major): one clip group of
    the temporal head at BASELINE config 1 size (5 frames x 100 slots)."""
    import torch
    from slotvps_amd import ops
    g = torch.Generator(device=cuda).manual_seed(11)
    G, Lq, C = 3, 500, 256
    q = torch.randn((G, Lq, C), generator=g, device=cuda)
    k = torch.randn((G, Lq, C), generator=g, device=cuda)
    v = torch.randn((G, Lq, C), generator=g, device=cuda)
    logits_t = ops.bgemm(k, q)                                   # [G, Lk, Lq]
    _check(logits_t, k, q, None, 1.0)
    attn_t = torch.softmax(logits_t, dim=-1)
    out = ops.bgemm(attn_t.transpose(1, 2), v.transpose(1, 2))   # A[m = lq, k = lk], B[n = c, k = lk]
    _check(out, attn_t.transpose(1, 2), v.transpose(1, 2), None, 1.0)


@pytest.mark.parametrize("split", ["fp16", "bf16"])
def test_bgemm_temporal_retriever_viper_clip(cuda, split):
    """The temporal slot retriever of BASELINE config 5 (T = 10 frames x 200 slots = 2000 rows, dynamic_mask_head.py:559-567) exactly as
    SlotsDynamicConv.forward runs it: logits^T = k q^T [2000, 2000], softmax over the QUERY axis (scaled by 2^14 in the fp16 split), then
    attn^T v with the probabilities read k-major - two clips per launch - against float64."""
    import torch
    from slotvps_amd import ops
    g = torch.Generator(device=cuda).manual_seed(13)
    G, Lq, C = 2, 2000, 256
    q, k, v = (torch.nn.functional.layer_norm(torch.randn((G, Lq, C), generator=g, device=cuda), (C,)) * 0.5 for _ in range(3))
    psc = 16384.0 if split == "fp16" else 1.0
    logits_t = ops.bgemm(k, q, split=split)
    want_l = torch.matmul(k.double(), q.double().transpose(1, 2))
    e_l = (logits_t.double() - want_l).abs().max().item() / want_l.abs().max().item()
    attn_t = ops.row_softmax(logits_t, inplace=False, scale=psc)
    out = ops.bgemm(attn_t.transpose(1, 2), v.transpose(1, 2), split=split, alpha=1.0 / psc)
    want = torch.matmul(torch.softmax(want_l, dim=-1).transpose(1, 2), v.double())
    e_o = (out.double() - want).abs().max().item() / want.abs().max().item()
    print(f"temporal retriever at T L = 2000, split {split}: logits {e_l:.2e}, output {e_o:.2e} (relative to the largest value)")
    assert e_l <= (1e-6 if split == "fp16" else 4e-5) and e_o <= (2e-5 if split == "fp16" else 2e-4), (e_l, e_o)


def test_bgemm_refuses_cpu_and_bad_shapes(cuda):
    import torch
    from slotvps_amd import ops
    with pytest.raises(RuntimeError):
        ops.bgemm(torch.zeros(2, 3), torch.zeros(4, 3))
    with pytest.raises(ValueError):
        ops.bgemm(torch.zeros((2, 3, 4), device=cuda), torch.zeros((2, 5, 6), device=cuda))
    with pytest.raises(ValueError):
        ops.bgemm(torch.zeros((2, 3, 4), device=cuda), torch.zeros((3, 5, 4), device=cuda))
