"""Host-side constants of the statistics-fused retriever (no GPU): the QR factors reproduce the LayerNorm variance, the
projected position tables are the key factor applied to the separable sine tables (slot_head.MaskDynamicConv)."""
import torch

from slotvps_amd.slot_head import MaskDynamicConv


def test_qr_factor_gives_the_layernorm_variance_and_the_projected_tables():
    torch.manual_seed(3)
    m = MaskDynamicConv(256).double()
    c = MaskDynamicConv._fused_consts(m)
    rk = c["rk64"]                                                  # [256, 256] upper triangular, float64
    assert torch.equal(rk, torch.triu(rk))
    x = torch.randn(50, 256, dtype=torch.float64)
    u = m.to_k(x)
    var = u.var(dim=1, unbiased=False)
    rb = c["rbk"].double().cpu()
    got = ((x @ rk.t() + rb) ** 2).sum(1) / 256.0
    assert torch.allclose(got, var, rtol=1e-6, atol=1e-9)           # |R x + r|^2 / 256 = var(W x + b) (rb is stored as fp32)
    # fp16 factor handed to the kernel: upper triangular as well, within fp16 rounding of the float64 one
    assert torch.equal(c["rk"].cpu(), torch.triu(c["rk"].cpu()))
    assert (c["rk"].double().cpu() - rk).abs().max() <= 2.0 ** -11 * rk.abs().max()
    # projected position tables: R_k (f + pos) = R_k f + ty[y] + tx[x] for separable pos = [ytab[y] | xtab[x]]
    H, W = 5, 7
    ytab = torch.randn(H, 128)
    xtab = torch.randn(W, 128)
    ty, tx = m.retr_pos_tables((ytab, xtab))
    assert ty.shape == (H, 256) and tx.shape == (W, 256) and ty.dtype == torch.float32
    y, xx = 3, 6
    pos = torch.cat([ytab[y], xtab[xx]]).double()
    want = rk @ pos
    assert torch.allclose(ty[y].double() + tx[xx].double(), want, rtol=1e-5, atol=1e-5)
    # cached per (weights, geometry): the same objects come back
    ty2, tx2 = m.retr_pos_tables((ytab, xtab))
    assert ty2 is ty and tx2 is tx
