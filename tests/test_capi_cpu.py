"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every
symbol the header declares, the ctypes table covers the header, and the product never imports the oracle."""
import os
import re

from util import ROOT


def _header_symbols(name="slotvps_hip.h"):
    text = open(os.path.join(ROOT, "include", name)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"typedef[^;]*;", "", text)                       # (the launch-hook function-pointer type is not an entry point)
    return sorted(set(re.findall(r"\b(svps_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from slotvps_amd import _lib
    lib = _lib.load()                       # raises if the .so is missing or a symbol is absent
    syms = _header_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/slotvps_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == syms, (sorted(set(syms) ^ set(_lib.SIGNATURES)))
    assert lib.svps_abi_version() == 1


def test_product_library_carries_no_diagnostics():
    """the product library exports exactly the symbols of include/slotvps_hip.h - no probes, no event bookkeeping, no
    stamp / ablation code; the diagnostics live in libslotvps_hip_diag.so behind include/slotvps_hip_diag.h and reach the product only
    through its launch hook."""
    import subprocess
    from slotvps_amd import _lib

    def exported(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        return sorted(ln.split()[-1] for ln in out.splitlines() if " T " in ln and ln.split()[-1].startswith("svps_"))
    prod = exported(_lib.LIB_PATH)
    assert prod == _header_symbols(), sorted(set(prod) ^ set(_header_symbols()))
    assert not [s for s in prod if "probe" in s or "stamp" in s or "debug" in s or s in ("svps_prof_enable", "svps_prof_collect", "svps_prof_reset")]
    diag = exported(_lib.DIAG_LIB_PATH)
    assert diag == _header_symbols("slotvps_hip_diag.h") == sorted(_lib.DIAG_SIGNATURES), sorted(set(diag) ^ set(_lib.DIAG_SIGNATURES))
    assert not set(diag) & set(prod)
    lib, d = _lib.load(), _lib.load_diag()                       # loads without a GPU; installing the hook touches no device
    d.svps_prof_enable(0)
    lib.svps_prof_mark(0, 0, None)                               # recorder disabled: the hook returns at once


def test_argument_errors_do_not_need_a_gpu():
    """Entry points validate before touching the device: bad arguments return SVPS_ERR_* codes."""
    import ctypes
    from slotvps_amd import _lib
    lib = _lib.load()
    assert lib.svps_slot_attn_workspace_bytes(0, 100, 10, 0) == 0
    assert lib.svps_slot_attn_fwd(None, None, None, None, None, 1e-5, None, 0, None, None, 1, 100, 64, 256, 0, 0, None) == -1
    assert lib.svps_mask_decode_fwd(None, None, None, None, 1.0, 0.0, None, None, 1, 1, 1, 256, 0, None) == -1
    assert lib.svps_pos_embed_sine(None, 4, 4, 256, None) == -1
    one = ctypes.c_void_p(16)                     # never dereferenced: shape errors are reported first
    big = (1 << 22) + 1                           # frames above 4 Mi pixels: 32-bit offsets inside a frame
    assert lib.svps_slot_attn_fwd(one, one, one, one, one, 1e-5, one, 1 << 40, one, None, 1, 100, big, 256, 1, 0, None) == -2
    assert lib.svps_slot_attn_fwd(one, one, one, one, one, 1e-5, one, 1 << 40, one, None, 1, 257, 64, 256, 1, 0, None) == -2
    assert lib.svps_slot_attn_fwd(one, one, one, one, one, 1e-5, one, 16, one, None, 1, 100, 4096, 256, 1, 0, None) == -3
    assert lib.svps_mask_decode_fwd(one, one, one, one, 1.0, 0.0, one, None, 1, 100, big, 256, 0, None) == -2
    # the round-2 entry points: statistics-fused retriever, K8 with the LayerNorm epilogue, K9
    assert lib.svps_retr_stats_fwd(None, None, None, None, None, 1e-5, None, None, 1e-5, None, 1, 4, 32, 256, 0, None) == -1
    assert lib.svps_retr_stats_fwd(one, one, None, one, one, 1e-5, one, one, 1e-5, one, 1, 4, 32, 256, 0, None) == -1   # ty without tx
    assert lib.svps_retr_stats_fwd(one, None, None, one, one, 1e-5, one, one, 1e-5, one, 1, 4, 32, 128, 0, None) == -2  # D != 256
    assert lib.svps_retr_attn_fwd(one, one, one, one, one, one, None, one, 1 << 30, one, 1, 100, 4, 32, 256, 0, 0, None) == -1
    assert lib.svps_retr_attn_fwd(one, one, one, one, one, one, one, one, 1 << 30, one, 1, 257, 4, 32, 256, 0, 0, None) == -2
    assert lib.svps_retr_attn_fwd(one, one, one, one, one, one, one, one, 16, one, 1, 100, 4, 32, 256, 0, 2, None) == -3   # workspace too small (flags: fp16 map)
    assert lib.svps_retr_attn_workspace_bytes(1, 100, 4, 32, 0) > 0 and lib.svps_retr_attn_workspace_bytes(0, 100, 4, 32, 0) == 0
    # round 6: the multi-order level fusion (host arrays of device pointers) and the reference-precision entries validate before any launch
    arr = (ctypes.c_void_p * 2)(16, 16)
    assert lib.svps_level_fuse_hl_multi_fwd(None, None, 2, arr, arr, arr, arr, one, one, arr, 1, 8, 32, None) == -1          # no incoming map
    assert lib.svps_level_fuse_hl_multi_fwd(one, None, 5, arr, arr, arr, arr, one, one, arr, 1, 8, 32, None) == -1           # 1 ... 4 orders
    assert lib.svps_level_fuse_hl_multi_fwd(one, None, 2, arr, arr, arr, arr, one, one, arr, 1, 7, 32, None) == -2           # x2 upsampling: even sizes
    assert lib.svps_level_fuse_hl_fwd(one, one, one, one, one, one, None, None, 1, 8, 32, None) == -1                       # planes: both or neither
    assert lib.svps_retr_attn_hl_fwd(one, one, one, one, one, one, one, one, one, 16, one, 1, 257, 4, 32, 256, 0, None) == -2
    assert lib.svps_mask_decode_hl_fwd(one, one, one, one, one, 1.0, 0.0, None, None, 1, 100, 64, 256, None) == -1           # fp32 logits are required
    assert lib.svps_slot_gemm_ln(one, one, None, None, None, None, one, 1e-5, 0, one, 8, 256, None) == -1             # no gamma
    assert lib.svps_slot_gemm_ln(one, one, None, None, None, one, one, 1e-5, 0, one, 8, 250, None) == -2              # K % 16
    three = (ctypes.c_longlong * 3)(1, 1, 1)
    assert lib.svps_bgemm(one, three, one, three, None, None, one, None, 1, 4, 4, 4, 1.0, None) == -1                 # no C strides
    assert lib.svps_bgemm(one, three, one, three, None, None, one, three, 0, 4, 4, 4, 1.0, None) == -2                # empty batch
    assert lib.svps_bgemm(one, three, one, three, one, None, one, three, 1, 4, 4, 4, 1.0, None) == -1                 # bias without strides
    c, t, p = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert lib.svps_slot_attn_plan(5, 100, 131072, 0, ctypes.byref(c), ctypes.byref(t), ctypes.byref(p)) == 0
    assert c.value * t.value * p.value >= 131072 and (c.value - 1) * t.value * p.value < 131072


def test_product_never_imports_the_oracle():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|slotvps_oracle", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "slotvps_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not pat.search(text), f"{f} references the oracle"


def test_ops_fail_loudly_without_gpu_tensors():
    import pytest
    import torch
    from slotvps_amd import ops
    x = torch.zeros((1, 8, 256), dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):
        ops.slot_attn(x, x, x, torch.ones(256), torch.zeros(256))
    with pytest.raises(RuntimeError):
        ops.pos_embed_sine(4, 4, 256, "cpu")
    with pytest.raises(RuntimeError):
        ops.level_fuse(torch.zeros(1, 128, 2, 2), None, torch.zeros(256, 384, dtype=torch.bfloat16), torch.zeros(256), 2, 2)
