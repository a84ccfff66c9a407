"""Robustness of the fp16 staging of the fused retriever (K3' / K3'' + K1'): the stored bf16 map is converted to fp16 in LDS
(exact for |f| in [6.1e-5, 65504]) and Q'' is carried as fp16 hi + lo. Feature maps of trained checkpoints are not N(0, 1) and
LayerNorm affines are not U(0.5, 1.5): these cases scale the map over six decades, stretch gamma / beta, and degenerate the
variance, always against a float64 evaluation of MaskDynamicConv.forward (dynamic_mask_head.py:423-461) on the same bf16 map.
Above the fp16 range the documented behaviour is saturation; `range_check` routes such maps to the kv form (another HIP kernel
pair - never the oracle)."""
import warnings

import numpy as np
import pytest

from util import orc, to_bf16_t
from test_retr_fused_gpu import make_module

pytestmark = pytest.mark.gpu


def run_case(cuda, m, P, feat, slots, H, W, pos=True):
    import torch
    from slotvps_amd import ops
    tabs = ops.pos_embed_sine_tables(H, W, 256, cuda) if pos else None
    with torch.no_grad():
        got = m.forward_pm(torch.from_numpy(slots).to(cuda), to_bf16_t(feat, cuda), (H, W), tabs).cpu().numpy()
    pm = orc.pos_embed_sine(H, W) if pos else None
    worst = 0.0
    for t in range(feat.shape[0]):
        ref = orc.retriever(slots[t], feat[t], pm, P, "", st=orc.Storage.exact(), dt=np.float64)
        worst = max(worst, float(np.abs(got[t] - ref).max()))
    assert np.isfinite(got).all()
    return worst


@pytest.mark.parametrize("scale", [1e-6, 1e-3, 1.0, 30.0, 1e3])
def test_map_magnitude(cuda, scale):
    """|f| from 1e-6 (fp16 subnormals: the bits lost there sit 1e5 below the bias term that then dominates the projections) to
    1e3 (exact in fp16; rstd_v ~ 1e-3 pushes P * rstd_v towards the fp16 subnormals). Outputs are post-LayerNorm, O(1)."""
    m, P = make_module(cuda, 41)
    rng = np.random.default_rng(5)
    T, H, W, L = 1, 12, 40, 100
    feat = orc.round_bf16((scale * rng.standard_normal((T, H * W, 256))).astype(np.float32))
    slots = rng.standard_normal((T, L, 256)).astype(np.float32)
    err = run_case(cuda, m, P, feat, slots, H, W)
    print(f"\nmap scale {scale:g}: max abs err vs float64 oracle {err:.2e}")
    assert err <= 3e-3


@pytest.mark.parametrize("glo,ghi", [(0.01, 0.05), (5.0, 20.0), (0.01, 20.0)])
def test_layernorm_affine_range(cuda, glo, ghi):
    """gamma in [0.01, 20] on all four LayerNorms, beta ~ N(0, 1): logits reach +-1e3 (gamma_q gamma_k = 400), far past the range
    the max-subtracted softmax is exercised with at N(0, 1) weights."""
    import torch
    m, P = make_module(cuda, 43)
    rng = np.random.default_rng(7)
    with torch.no_grad():
        for n in ("norm_q", "norm_k", "norm_v", "norm1"):
            P[f"{n}.weight"] = np.exp(rng.uniform(np.log(glo), np.log(ghi), 256)).astype(np.float32)
            P[f"{n}.bias"] = rng.standard_normal(256).astype(np.float32)
            getattr(m, n).weight.copy_(torch.from_numpy(P[f"{n}.weight"]))
            getattr(m, n).bias.copy_(torch.from_numpy(P[f"{n}.bias"]))
    T, H, W, L = 1, 12, 40, 100
    feat = orc.round_bf16(rng.standard_normal((T, H * W, 256)).astype(np.float32))
    slots = rng.standard_normal((T, L, 256)).astype(np.float32)
    err = run_case(cuda, m, P, feat, slots, H, W)
    scale = max(1.0, float(np.abs(P["norm1.weight"]).max()))
    print(f"\ngamma in [{glo}, {ghi}]: max abs err vs float64 oracle {err:.2e} (output scale {scale:.1f})")
    assert err <= 3e-3 * scale


def test_zero_variance_pixels(cuda):
    """A zero map with zero projection biases: W x + b = 0 on the value side for every pixel (variance 0 -> rstd = 1 / sqrt(eps));
    the key side sees the position embedding only."""
    import torch
    m, P = make_module(cuda, 47)
    with torch.no_grad():
        for n in ("to_k", "to_v"):
            P[f"{n}.bias"] = np.zeros(256, dtype=np.float32)
            getattr(m, n).bias.zero_()
    rng = np.random.default_rng(9)
    T, H, W, L = 1, 8, 32, 50
    feat = np.zeros((T, H * W, 256), dtype=np.float32)
    slots = rng.standard_normal((T, L, 256)).astype(np.float32)
    err = run_case(cuda, m, P, feat, slots, H, W)
    print(f"\nzero map, zero biases: max abs err vs float64 oracle {err:.2e}")
    assert err <= 3e-3


def test_map_beyond_fp16_range_falls_back_to_kv_form(cuda):
    """|f| up to 3e5 > 65504: with range_check the module runs the kv form (bf16 keys and values: 1e-1-class on O(1)
    outputs, tests/test_head_gpu.py) and warns once; without it the fused form saturates - finite, documented, not asserted close."""
    import torch
    from slotvps_amd.slot_head import MaskDynamicConv
    m, P = make_module(cuda, 53)
    rng = np.random.default_rng(11)
    T, H, W, L = 1, 8, 32, 100
    feat = orc.round_bf16((1e5 * rng.standard_normal((T, H * W, 256))).astype(np.float32))
    slots = rng.standard_normal((T, L, 256)).astype(np.float32)
    m.range_check = True
    MaskDynamicConv._warned_range = False
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        err = run_case(cuda, m, P, feat, slots, H, W)
    assert any("fp16 range" in str(w_.message) for w_ in rec)
    print(f"\n|f| ~ 1e5 with range_check (kv form): max abs err vs float64 oracle {err:.2e}")
    assert err <= 6e-1                                           # (measured 3.9e-1: every key / value rounded to bf16 at |f| ~ 1e5)
    m.range_check = False
    sat = run_case(cuda, m, P, feat, slots, H, W)                # saturating fused form: finite (asserted inside), error reported
    print(f"|f| ~ 1e5 without range_check (fp16 saturation at 65504): max abs err {sat:.2e}")



def test_tiny_norm_v_eps_is_refused(cuda):
    """The fp16 probabilities carry 2^7 (csrc/common.h; slots that own almost no pixel - P ~ 1e-7 everywhere - no longer fall below
    fp16's range: tests/test_head_gpu.py, the fine levels of the fixture). The guard that goes with the scale: 2^7 * P * rstd_v must
    stay inside fp16, rstd_v <= 1 / sqrt(eps_v) < 511, so the fused form refuses eps_v < 4e-6 instead of overflowing."""
    m, P = make_module(cuda, 61)
    rng = np.random.default_rng(13)
    T, H, W, L = 1, 8, 32, 100
    feat = orc.round_bf16(rng.standard_normal((T, H * W, 256)).astype(np.float32))
    slots = rng.standard_normal((T, L, 256)).astype(np.float32)
    assert run_case(cuda, m, P, feat, slots, H, W) <= 2e-3
    m.norm_v.eps = 1e-6
    with pytest.raises(ValueError):
        run_case(cuda, m, P, feat, slots, H, W)
    m.norm_v.eps = 1e-5
