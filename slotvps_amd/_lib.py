"""ctypes binding of libslotvps_hip.so (C ABI declared in include/slotvps_hip.h).

The library is the product: there is no Python/NumPy/torch fallback for any entry point. If the
shared object is missing or a symbol cannot be resolved the import of the calling op raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SLOTVPS_LIB") or os.path.join(_HERE, "libslotvps_hip.so")   # env: A/B builds
# diagnostics (include/slotvps_hip_diag.h): HIP-event timing behind the product's launch hook, hardware / streaming probes. Loaded on
# demand by bench.py's roofline leg, tests/test_probes_gpu.py and tools/ (load_diag); nothing in this package needs it to run
DIAG_LIB_PATH = os.environ.get("SLOTVPS_DIAG_LIB") or os.path.join(_HERE, "libslotvps_hip_diag.so")

ERR_NAMES = {-1: "SVPS_ERR_BAD_ARG", -2: "SVPS_ERR_BAD_SHAPE", -3: "SVPS_ERR_WORKSPACE"}

FLAG_SPLIT_P = 1
FLAG_OUT_BF16 = 1
FLAG_MAP_F16 = 2          # the fused map is fp16, not bf16 (SVPS_FLAG_MAP_F16)

KERNEL_SLOT_ATTN = 0
KERNEL_SLOT_ATTN_FINISH = 1
KERNEL_MASK_DECODE = 2
KERNEL_POS_EMBED = 3
KERNEL_KV_PROJECT = 4
KERNEL_LEVEL_FUSE = 5
KERNEL_PANOPTIC_POST = 6
KERNEL_DEFORM_CONV = 7
KERNEL_RETR_STATS = 8
KERNEL_RETR_ATTN = 9
KERNEL_RETR_FINISH = 10

_c = ctypes
_vp, _i, _f, _sz = _c.c_void_p, _c.c_int, _c.c_float, _c.c_size_t

# name -> (restype, argtypes); must list every symbol of include/slotvps_hip.h
SIGNATURES = {
    "svps_abi_version": (_i, []),
    "svps_slot_attn_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "svps_slot_attn_plan": (_i, [_i, _i, _i, _i, _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i)]),
    "svps_slot_attn_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _vp, _sz, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "svps_mask_decode_fwd": (_i, [_vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "svps_pos_embed_sine": (_i, [_vp, _i, _i, _i, _vp]),
    "svps_pos_embed_sine_tables": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "svps_kv_project_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _f, _vp, _vp,
                                 _i, _i, _i, _i, _vp]),
    "svps_level_fuse_fwd": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "svps_row_ln": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _vp, _vp, _vp]),
    "svps_row_softmax": (_i, [_vp, _vp, _i, _i, _vp]),
    "svps_row_softmax_scaled": (_i, [_vp, _vp, _i, _i, _f, _vp]),
    "svps_group_norm_relu_workspace_bytes": (_sz, [_i, _i, _i]),
    "svps_group_norm_relu_fwd": (_i, [_vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _sz, _i, _i, _i, _vp]),
    "svps_group_norm_relu16_fwd": (_i, [_vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _i, _vp, _sz, _i, _i, _i, _vp]),
    "svps_group_norm_relu_stats_fwd": (_i, [_vp, _vp, _i, _vp, _vp, _i, _f, _vp, _vp, _vp, _i, _vp, _sz, _i, _i, _i, _vp]),
    "svps_nchw_to_pixel_major": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "svps_semantic_pred_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "svps_conv3x3_pm_small_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "svps_deform_conv_fused_stats_chunks": (_i, [_i, _i, _i, _i]),
    "svps_deform_conv_fused_stats_fwd": (_i, [_vp, _vp, _vp, _vp, _vp] + [_i] * 12 + [_vp]),
    "svps_retr_query_prep": (_i, [_vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "svps_retr_split": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "svps_slot_self_attn": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "svps_slot_self_attn_f16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "svps_panoptic_candidates": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "svps_panoptic_argmax": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "svps_panoptic_clip_state_ints": (_i, []),
    "svps_panoptic_clip_select": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp]),
    "svps_panoptic_clip": (_i, [_vp, _c.c_longlong, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _f, _c.c_double, _i, _i, _i, _i, _vp]),
    "svps_deform_im2col": (_i, [_vp, _vp, _vp] + [_i] * 15 + [_vp]),
    "svps_deform_im2col_bf16": (_i, [_vp, _vp, _vp] + [_i] * 15 + [_vp]),
    "svps_slot_gemm_ln": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _vp, _i, _i, _vp]),
    "svps_slot_chain": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "svps_slot_ffn": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _vp, _i, _i, _vp]),
    "svps_slot_ffn_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _vp, _i, _i, _vp]),
    "svps_bgemm": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "svps_bgemm_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "svps_retr_stats_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _f, _vp, _i, _i, _i, _i, _i, _vp]),
    "svps_retr_stats_level_fwd": (_i, [_vp, _i] + [_vp] * 9 + [_i, _i, _i, _i, _i, _vp]),
    "svps_retr_attn_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "svps_retr_attn_fwd": (_i, [_vp] * 8 + [_sz, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "svps_level_fuse_hl_fwd": (_i, [_vp] * 8 + [_i, _i, _i, _vp]),   # cur, gprev, wb_hi, wb_lo, bc, out_hi, out_lo, out_f32
    "svps_level_fuse_hl_pm_fwd": (_i, [_vp] * 9 + [_i, _i, _i, _vp]),   # cur_hi, cur_lo, gprev, wb_hi, wb_lo, bc, out_hi, out_lo, out_f32
    "svps_level_fuse_hl_multi_fwd": (_i, [_vp, _vp, _i] + [_vp] * 7 + [_i, _i, _i, _vp]),   # cur, cur_lo, n, gprev[], wb_hi[], wb_lo[], bc[], out_hi, out_lo, out_f32[]
    "svps_retr_stats_hl_fwd": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _vp, _vp, _f, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _i, _vp]),
    "svps_retr_attn_hl_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "svps_retr_attn_hl_fwd": (_i, [_vp] * 9 + [_sz, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "svps_mask_decode_hl_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _i, _i, _i, _i, _vp]),
    "svps_level_fuse_f32_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "svps_kv_project_f32_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _f, _vp, _vp,
                                     _i, _i, _i, _i, _vp]),
    "svps_slot_attn_f32_workspace_bytes": (_sz, [_i, _i, _i]),
    "svps_slot_attn_f32_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _vp, _sz, _vp, _vp, _i, _i, _i, _i, _vp]),
    "svps_mask_decode_f32_fwd": (_i, [_vp, _vp, _vp, _vp, _f, _f, _vp, _i, _i, _i, _i, _vp]),
    "svps_deform_conv_fused_fwd": (_i, [_vp, _vp, _vp, _vp] + [_i] * 12 + [_vp]),
    "svps_slot_gemm": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "svps_slot_gemm_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "svps_slot_gemm_f16_act": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "svps_slot_gemm_ln_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _vp, _i, _i, _vp]),
    "svps_set_launch_hook": (None, [_vp]),
    "svps_prof_mark": (None, [_i, _i, _vp]),
}

# name -> (restype, argtypes); must list every symbol of include/slotvps_hip_diag.h
DIAG_SIGNATURES = {
    "svps_diag_launch_hook": (None, [_i, _i, _vp]),
    "svps_prof_enable": (None, [_i]),
    "svps_prof_reset": (None, []),
    "svps_prof_collect": (_i, [_i, _c.POINTER(_c.c_double), _c.POINTER(_i)]),
    "svps_probe_mfma": (_i, [_vp, _vp, _vp, _vp]),
    "svps_probe_tile": (_i, [_vp, _vp, _vp, _vp]),
    "svps_probe_copy": (_i, [_vp, _vp, _sz, _vp]),
    "svps_probe_mix": (_i, [_vp, _vp, _sz, _i, _i, _vp]),
    "svps_probe_mfma_feed": (_i, [_i, _i, _i, _i, _vp, _vp, _vp]),
    "svps_probe_cvt_fp8": (_i, [_vp, _f, _vp, _i, _vp]),
    "svps_probe_mx_block": (_i, [_vp] * 7),
    "svps_probe_mx_fp8": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp]),
}

_lib = None
_diag = None


class SlotVPSLibraryError(RuntimeError):
    pass


def load():
    """Load the HIP library once and attach prototypes. Raises SlotVPSLibraryError if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SlotVPSLibraryError(
            f"{LIB_PATH} not found: build it with `make -C slotvps_amd/csrc` or "
            "`python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU fallback.")
    # PyTorch-ROCm first: the library's libamdhip64 dependency must resolve to the HIP runtime torch has loaded. Loaded before torch, it
    # binds /opt/rocm's copy, torch then brings its own, and the second ROCr instance of the process finds no device (hipError 100 at
    # the first kernel: seen with `python __graft_entry__.py smoke`, where build() loaded the library before anything imported torch)
    import torch  # noqa: F401
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 missing
        raise SlotVPSLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise SlotVPSLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    ver = lib.svps_abi_version()
    if ver != 1:
        raise SlotVPSLibraryError(f"ABI version {ver} != 1")
    _lib = lib
    return lib


def load_diag():
    """Load the diagnostics library once (include/slotvps_hip_diag.h) and install its HIP-event recorder as the product's launch hook
    (svps_set_launch_hook): recording itself stays off until svps_prof_enable(1) (ops.KernelTimer). Raises if it is absent - a
    measurement must not silently measure nothing."""
    global _diag
    if _diag is not None:
        return _diag
    lib = load()
    if not os.path.exists(DIAG_LIB_PATH):
        raise SlotVPSLibraryError(f"{DIAG_LIB_PATH} not found: build it with `make -C slotvps_amd/csrc`")
    try:
        diag = ctypes.CDLL(DIAG_LIB_PATH)
    except OSError as e:
        raise SlotVPSLibraryError(f"cannot load {DIAG_LIB_PATH}: {e}") from e
    for name, (res, args) in DIAG_SIGNATURES.items():
        try:
            fn = getattr(diag, name)
        except AttributeError as e:
            raise SlotVPSLibraryError(f"{DIAG_LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    lib.svps_set_launch_hook(ctypes.cast(diag.svps_diag_launch_hook, ctypes.c_void_p))
    _diag = diag
    return diag


def check(code, what):
    if code == 0:
        return
    if code < 0:
        raise ValueError(f"{what}: {ERR_NAMES.get(code, code)}")
    raise RuntimeError(f"{what}: hipError_t {code}")
