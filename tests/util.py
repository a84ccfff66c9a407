"""Helpers shared by the tests (the oracle is imported here and in tests only)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import slotvps_oracle as orc  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def ln_like(rng, n, d=256):
    """LayerNorm-shaped rows with a non-trivial affine (w~U(0.5,1.5), b~N(0,0.1))."""
    x = rng.standard_normal((n, d)).astype(np.float32)
    w = rng.uniform(0.5, 1.5, d).astype(np.float32)
    b = (0.1 * rng.standard_normal(d)).astype(np.float32)
    return orc.layer_norm(x, w, b).astype(np.float32)


def to_bf16_t(x, device):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(device).to(torch.bfloat16).contiguous()


def bf16_t_to_np(t):
    return t.float().cpu().numpy()
