"""Pins the gfx950 instruction semantics the kernels rely on (MFMA lane maps, transposed LDS read,
swizzled LDS-DMA image) through the probes of the DIAGNOSTICS library (include/slotvps_hip_diag.h, libslotvps_hip_diag.so)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_probe_mfma_layout(cuda):
    import torch
    from slotvps_amd import _lib
    lib = _lib.load_diag()
    rng = np.random.default_rng(0)
    # small integers: exact in bf16 and in the fp32 accumulator; asymmetric operands
    a = rng.integers(-4, 5, (32, 16)).astype(np.float32)
    b = rng.integers(-4, 5, (16, 32)).astype(np.float32)
    ta = torch.from_numpy(a).to(cuda).to(torch.bfloat16).contiguous()
    tb = torch.from_numpy(b).to(cuda).to(torch.bfloat16).contiguous()
    tc = torch.zeros((32, 32), dtype=torch.float32, device=cuda)
    rc = lib.svps_probe_mfma(ctypes.c_void_p(ta.data_ptr()), ctypes.c_void_p(tb.data_ptr()),
                             ctypes.c_void_p(tc.data_ptr()), ctypes.c_void_p(0))
    assert rc == 0
    torch.cuda.synchronize()
    np.testing.assert_array_equal(tc.cpu().numpy(), a @ b)


def test_probe_tile_roundtrip(cuda):
    import torch
    from slotvps_amd import _lib
    lib = _lib.load_diag()
    x = (np.arange(32 * 256, dtype=np.float32).reshape(32, 256) % 251) - 125  # exact in bf16
    x += (np.arange(32, dtype=np.float32)[:, None] % 3)
    tx = torch.from_numpy(x).to(cuda).to(torch.bfloat16).contiguous()
    rows = torch.full((32, 256), -7.0, dtype=torch.bfloat16, device=cuda)
    cols = torch.full((32, 256), -7.0, dtype=torch.bfloat16, device=cuda)
    rc = lib.svps_probe_tile(ctypes.c_void_p(tx.data_ptr()), ctypes.c_void_p(rows.data_ptr()),
                             ctypes.c_void_p(cols.data_ptr()), ctypes.c_void_p(0))
    assert rc == 0
    torch.cuda.synchronize()
    ref = tx.float().cpu().numpy()
    got_rows = rows.float().cpu().numpy()
    got_cols = cols.float().cpu().numpy()
    bad_r = np.argwhere(got_rows != ref)
    assert bad_r.size == 0, f"row-fragment read wrong at {bad_r[:8].tolist()} (of {len(bad_r)})"
    bad_c = np.argwhere(got_cols != ref)
    assert bad_c.size == 0, f"transposed read wrong at {bad_c[:8].tolist()} (of {len(bad_c)}): got {got_cols[tuple(bad_c[0])]} want {ref[tuple(bad_c[0])]}"
