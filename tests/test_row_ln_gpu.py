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
