"""Tensor-level entry points of the HIP path (thin: argument checks + one C-ABI call each).

PyTorch is used for device memory and the current stream only. Every function requires CUDA
(ROCm) tensors and the built libslotvps_hip.so; there is deliberately no CPU path here - the CPU
restatement used for checking lives in oracle/ and is never imported from this package.
"""
import ctypes

import torch

from . import _lib

D_MODEL = 256


class _on:
    """Launch context of one C-ABI call: all tensor arguments must live on ONE GPU; that GPU is made current for the
    call (kernel attributes, CU count and the launch itself are per device) and `.stream` is the current stream OF
    THAT DEVICE - not of whichever device happens to be current in the caller."""

    def __init__(self, *tensors):
        devs = {t.device for t in tensors if isinstance(t, torch.Tensor)}
        if len(devs) != 1:
            raise ValueError(f"tensor arguments on different devices: {sorted(map(str, devs))}")
        self.device = devs.pop()
        if self.device.type != "cuda":
            raise RuntimeError(f"GPU tensors only (got {self.device}); there is no CPU fallback")
        self._guard = None

    def __enter__(self):
        if self.device.index != torch.cuda.current_device():
            self._guard = torch.cuda.device(self.device)
            self._guard.__enter__()
        self.stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return self

    def __exit__(self, *exc):
        if self._guard is not None:
            self._guard.__exit__(*exc)
        return False


def _stream_ptr(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _map_flag(feat, name="feat", ndim=3):
    """The fused level maps are 16-bit: bf16 (default) or fp16 (MultiScaleDynamicMaskHead.map_dtype = "fp16": three more mantissa
    bits in the same bytes, |f| < 65 504). Returns the C flag (0 / SVPS_FLAG_MAP_F16) after the usual checks."""
    if isinstance(feat, torch.Tensor) and feat.dtype == torch.float16:
        _need(feat, name, torch.float16, ndim)
        return _lib.FLAG_MAP_F16
    _need(feat, name, torch.bfloat16, ndim)
    return 0


def _need(t, name, dtype, ndim=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: the slot-retriever ops run on the GPU only (got "
                           f"{'a non-tensor' if not isinstance(t, torch.Tensor) else t.device}); "
                           "there is no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    if ndim is not None and t.dim() != ndim:
        raise ValueError(f"{name}: expected {ndim} dims, got {tuple(t.shape)}")


def slot_attn_plan(T, L, HW, chunks=0):
    """(workgroups per frame, tiles per workgroup, pixels per tile) the launcher will use."""
    lib = _lib.load()
    c, tpc, tp = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(lib.svps_slot_attn_plan(T, L, HW, chunks, ctypes.byref(c), ctypes.byref(tpc), ctypes.byref(tp)),
               "svps_slot_attn_plan")
    return c.value, tpc.value, tp.value


def slot_attn(q, k, v, ln_w, ln_b, eps=1e-5, split_p=True, chunks=0, return_pre_ln=False):
    """K1: out[t] = ReLU(LN(softmax_over_slots(q[t] k[t]^T) v[t])).

    q [T, L, 256] bf16, k/v [T, HW, 256] bf16 (pixel-major), ln_w/ln_b [256] fp32.
    Returns out [T, L, 256] fp32 (and the pre-LayerNorm pixel sum if return_pre_ln).
    Mirrors MaskDynamicConv.forward lines 435-459 of the reference's dynamic_mask_head.py.
    """
    lib = _lib.load()
    _need(q, "q", torch.bfloat16, 3)
    _need(k, "k", torch.bfloat16, 3)
    _need(v, "v", torch.bfloat16, 3)
    _need(ln_w, "ln_w", torch.float32, 1)
    _need(ln_b, "ln_b", torch.float32, 1)
    T, L, D = q.shape
    HW = k.shape[1]
    if k.shape != (T, HW, D) or v.shape != (T, HW, D) or ln_w.numel() != D or ln_b.numel() != D:
        raise ValueError(f"shape mismatch q{tuple(q.shape)} k{tuple(k.shape)} v{tuple(v.shape)}")
    ws_bytes = lib.svps_slot_attn_workspace_bytes(T, L, HW, chunks)
    ws = torch.empty(max(ws_bytes, 4) // 4, dtype=torch.float32, device=q.device)
    out = torch.empty((T, L, D), dtype=torch.float32, device=q.device)
    pre = torch.empty_like(out) if return_pre_ln else None
    flags = _lib.FLAG_SPLIT_P if split_p else 0
    with _on(q, k, v, ln_w, ln_b) as ctx:
        rc = lib.svps_slot_attn_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(ln_w), _ptr(ln_b), float(eps), _ptr(ws),
                                    ws_bytes, _ptr(out), _ptr(pre), T, L, HW, D, flags, chunks, ctx.stream)
    _lib.check(rc, "svps_slot_attn_fwd")
    return (out, pre) if return_pre_ln else out


def mask_decode(feat, embed, bn_scale, bn_shift, fg_scale, fg_shift, out_bf16=False, want_argmax=False, want_logits=True):
    """K2: mask logits [T, L, HW] from the finest fused feature map [T, HW, 256] bf16 and the
    last-stage slot embeddings [T, L, 256] fp32 (generate_final_outputs, vps_temporal_slots.py:144-160).
    want_logits=False (with want_argmax): only the per-pixel slot argmax [T, HW] uint8 is written (returns (None, amax))."""
    lib = _lib.load()
    mflag = _map_flag(feat)
    if mflag and out_bf16:
        raise ValueError("mask_decode: bf16 logits are not built for an fp16 map")
    _need(embed, "embed", torch.float32, 3)
    _need(bn_scale, "bn_scale", torch.float32, 1)
    _need(bn_shift, "bn_shift", torch.float32, 1)
    T, HW, D = feat.shape
    L = embed.shape[1]
    if embed.shape != (T, L, D) or bn_scale.numel() != D or bn_shift.numel() != D:
        raise ValueError("shape mismatch")
    if not want_logits and not want_argmax:
        raise ValueError("nothing to compute")
    out = torch.empty((T, L, HW), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=feat.device) if want_logits else None
    amax = torch.empty((T, HW), dtype=torch.uint8, device=feat.device) if want_argmax else None
    with _on(feat, embed, bn_scale, bn_shift) as ctx:
        rc = lib.svps_mask_decode_fwd(_ptr(feat), _ptr(embed), _ptr(bn_scale), _ptr(bn_shift), float(fg_scale),
                                      float(fg_shift), _ptr(out), _ptr(amax), T, L, HW, D,
                                      (_lib.FLAG_OUT_BF16 if out_bf16 else 0) | mflag, ctx.stream)
    _lib.check(rc, "svps_mask_decode_fwd")
    return (out, amax) if want_argmax else out


def pos_embed_sine(H, W, D=D_MODEL, device="cuda"):
    """Pixel-major [H*W, D] fp32 sine embedding (PositionEmbeddingSine, position_encoding.py:236-256)."""
    lib = _lib.load()
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("pos_embed_sine runs on the GPU only; there is no CPU fallback")
    out = torch.empty((H * W, D), dtype=torch.float32, device=dev)
    with _on(out) as ctx:
        _lib.check(lib.svps_pos_embed_sine(_ptr(out), H, W, D, ctx.stream), "svps_pos_embed_sine")
    return out


def pos_embed_sine_tables(H, W, D=D_MODEL, device="cuda"):
    """Separable form of the sine embedding: (ytab [H, D/2], xtab [W, D/2]) fp32."""
    lib = _lib.load()
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("pos_embed_sine_tables runs on the GPU only; there is no CPU fallback")
    ytab = torch.empty((H, D // 2), dtype=torch.float32, device=dev)
    xtab = torch.empty((W, D // 2), dtype=torch.float32, device=dev)
    with _on(ytab, xtab) as ctx:
        _lib.check(lib.svps_pos_embed_sine_tables(_ptr(ytab), _ptr(xtab), H, W, D, ctx.stream),
                   "svps_pos_embed_sine_tables")
    return ytab, xtab


def kv_project(feat, H, W, pos_tabs, wk, bk, lnk_w, lnk_b, lnk_eps, wv, bv, lnv_w, lnv_b, lnv_eps):
    """K3: k = bf16(LN_k(W_k bf16(feat + pos) + b_k)), v = bf16(LN_v(W_v feat + b_v)) for all frames.
    feat [T, H*W, 256] bf16; pos_tabs = (ytab, xtab) or None; wk / wv bf16 [256, 256]; biases and
    LayerNorm affines fp32.
    (MaskDynamicConv.forward lines 432-433 of the reference's dynamic_mask_head.py.)"""
    lib = _lib.load()
    _need(feat, "feat", torch.bfloat16, 3)
    T, HW, D = feat.shape
    if HW != H * W:
        raise ValueError("feat rows != H*W")
    for name, x in (("wk", wk), ("wv", wv)):
        _need(x, name, torch.bfloat16, 2)
    for name, x in (("bk", bk), ("lnk_w", lnk_w), ("lnk_b", lnk_b), ("bv", bv), ("lnv_w", lnv_w), ("lnv_b", lnv_b)):
        _need(x, name, torch.float32, 1)
    ytab = xtab = None
    if pos_tabs is not None:
        ytab, xtab = pos_tabs
        _need(ytab, "pos_y", torch.float32, 2)
        _need(xtab, "pos_x", torch.float32, 2)
        if ytab.shape != (H, D // 2) or xtab.shape != (W, D // 2):
            raise ValueError("pos tables do not match (H, W)")
    k = torch.empty_like(feat)
    v = torch.empty_like(feat)
    with _on(feat, ytab, xtab, wk, bk, lnk_w, lnk_b, wv, bv, lnv_w, lnv_b) as ctx:
        rc = lib.svps_kv_project_fwd(_ptr(feat), _ptr(ytab), _ptr(xtab), _ptr(wk), _ptr(bk), _ptr(lnk_w), _ptr(lnk_b),
                                     float(lnk_eps), _ptr(wv), _ptr(bv), _ptr(lnv_w), _ptr(lnv_b), float(lnv_eps),
                                     _ptr(k), _ptr(v), T, H, W, D, ctx.stream)
    _lib.check(rc, "svps_kv_project_fwd")
    return k, v


def level_fuse(cur, prev, wc, bc, H, W, bf16_values=False):
    """K4: fused level map [T, H*W, 256] bf16 = conv1x1(cat(bilinear_x2(prev), cur)) (+ level-0 form when
    prev is None). cur: [T, 128, H, W] fp32 (NCHW, the reference's layout) or [T, H*W, 128] 16-bit pixel-major in the element type of
    the conv's operands (= wc's dtype); prev: [T, (H/2)*(W/2), 256] bf16; wc [256, 384] bf16; bc [256] fp32.
    (MultiScaleDynamicMaskHead.forward lines 171-188 of the reference's dynamic_mask_head.py.)
    wc (and prev) fp16: the fp16 form - operands, previous level and result fp16.
    bf16_values (fp16 form only): every rounding point rounds to bf16 and the value is stored in the fp16 encoding - the bf16 storage
    policy, bit for bit above fp16's subnormal range, in the encoding the consumers' matrix instructions take directly."""
    lib = _lib.load()
    if not isinstance(cur, torch.Tensor) or not cur.is_cuda:
        raise RuntimeError("level_fuse: GPU tensors only; there is no CPU fallback")
    if cur.dtype == torch.float32:
        _need(cur, "cur", torch.float32, 4)
        T = cur.shape[0]
        if cur.shape != (T, 128, H, W):
            raise ValueError(f"cur {tuple(cur.shape)} != [T, 128, {H}, {W}]")
        nchw = 1
    else:
        _need(cur, "cur", wc.dtype, 3)
        T = cur.shape[0]
        if cur.shape != (T, H * W, 128):
            raise ValueError(f"cur {tuple(cur.shape)} != [T, {H * W}, 128]")
        nchw = 0
    # bf16_values: bf16 weights, fp16-ENCODED previous level and result (bf16 values); otherwise the maps have the weight's type
    mdt = torch.float16 if (wc.dtype == torch.float16 or bf16_values) else torch.bfloat16
    _need(wc, "wc", torch.bfloat16 if bf16_values else mdt, 2)
    _need(bc, "bc", torch.float32, 1)
    if wc.shape != (256, 384):
        raise ValueError("wc must be [256, 384]")
    if prev is not None:
        _need(prev, "prev", mdt, 3)
        if prev.shape != (T, (H // 2) * (W // 2), 256) or H % 2 or W % 2:
            raise ValueError(f"prev {tuple(prev.shape)} does not match an {H}x{W} level")
    out = torch.empty((T, H * W, 256), dtype=mdt, device=cur.device)
    with _on(cur, prev, wc, bc) as ctx:
        _lib.check(lib.svps_level_fuse_fwd(_ptr(cur), nchw | (2 if mdt == torch.float16 else 0) | (4 if bf16_values else 0), _ptr(prev), _ptr(wc), _ptr(bc),
                                           _ptr(out), T, H, W, ctx.stream), "svps_level_fuse_fwd")
    return out


def row_ln(x, w, b, eps=1e-5, pre=None, post=None, relu=False, rows_per_group=None, out_bf16=False):
    """K5: y = LN(x [+ pre]) * w[g] + b[g] (+ReLU) (+post); x [..., 256] fp32; w, b [256] or [G, 256]
    (group g = row // rows_per_group). Returns fp32, or bf16 when out_bf16."""
    lib = _lib.load()
    _need(x, "x", torch.float32)
    shape = x.shape
    rows = x.numel() // shape[-1]
    if shape[-1] != D_MODEL:
        raise ValueError("row_ln works on rows of 256 values")
    for name, tns in (("pre", pre), ("post", post)):
        if tns is not None:
            _need(tns, name, torch.float32)
            if tns.shape != shape:
                raise ValueError(f"{name} shape mismatch")
    _need(w, "w", torch.float32)
    _need(b, "b", torch.float32)
    groups = w.numel() // D_MODEL
    if rows_per_group is None:
        rows_per_group = rows if groups == 1 else -(-rows // groups)
    if groups * rows_per_group < rows or b.numel() != w.numel():
        raise ValueError("affine groups do not cover the rows")
    out = torch.empty(shape, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    with _on(x, pre, post, w, b) as ctx:
        rc = lib.svps_row_ln(_ptr(x), _ptr(pre), _ptr(post), _ptr(w), _ptr(b), float(eps), int(bool(relu)), rows,
                             rows_per_group, D_MODEL, _ptr(None if out_bf16 else out), _ptr(out if out_bf16 else None),
                             ctx.stream)
    _lib.check(rc, "svps_row_ln")
    return out


def row_softmax(x, inplace=False, scale=1.0):
    """Softmax over the last index of a contiguous fp32 tensor (csrc/row_ops.hip), one wavefront per row. scale (a power of two):
    scale * softmax(x), for a consumer that carries the probabilities as fp16 hi + lo (bgemm(split="fp16", alpha=1 / scale))."""
    lib = _lib.load()
    _need(x, "x", torch.float32)
    cols = x.shape[-1]
    rows = x.numel() // cols
    out = x if inplace else torch.empty_like(x)
    with _on(x, out) as ctx:
        if scale == 1.0:
            rc = lib.svps_row_softmax(_ptr(x), _ptr(out), rows, cols, ctx.stream)
        else:
            rc = lib.svps_row_softmax_scaled(_ptr(x), _ptr(out), rows, cols, float(scale), ctx.stream)
    _lib.check(rc, "svps_row_softmax")
    return out


def nchw_to_pixel_major(x):
    """[N, C, H, W] fp32 contiguous -> [N, H, W, C] fp32 (csrc/gn_relu.hip: 32-pixel tiles through LDS; the framework's permute +
    contiguous runs at a third of the copy rate)."""
    lib = _lib.load()
    _need(x, "x", torch.float32, 4)
    N, C, H, W = x.shape
    y = torch.empty((N, H, W, C), dtype=torch.float32, device=x.device)
    with _on(x, y) as ctx:
        _lib.check(lib.svps_nchw_to_pixel_major(_ptr(x), _ptr(y), N, C, H * W, ctx.stream), "svps_nchw_to_pixel_major")
    return y


def pack_conv3x3_small(weight):
    """nn.Conv2d weight [O <= 32, C, 3, 3] -> the packed [32, 9 C] matrix of svps_conv3x3_pm_small_fwd (k = tap C + c, rows >= O zero)."""
    O, C, kh, kw = weight.shape
    if (kh, kw) != (3, 3) or O > 32 or C % 32:
        raise ValueError("pack_conv3x3_small: [O <= 32, C % 32 == 0, 3, 3] expected")
    m = torch.zeros((32, 9 * C), dtype=torch.float32, device=weight.device)
    m[:O] = weight.detach().float().permute(0, 2, 3, 1).reshape(O, 9 * C)          # (o, ty, tx, c)
    return pack_b_fragments(m)


def conv3x3_pm_small(x_nhwc, wpack, bias, O):
    """3 x 3 convolution (stride 1, padding 1) with few output channels on pixel-major activations (csrc/offset_conv.hip): x_nhwc
    [N, H, W, C] fp32 -> [N, O, H, W] fp32; wpack = pack_conv3x3_small(weight). The offset convolution of DeformConvWithOffset."""
    lib = _lib.load()
    _need(x_nhwc, "x_nhwc", torch.float32, 4)
    N, H, W, C = x_nhwc.shape
    _need(wpack, "wpack", torch.bfloat16, 5)
    if wpack.shape[0] != 1 or wpack.shape[1] * 16 != 9 * C:
        raise ValueError("conv3x3_pm_small: packed weight does not match C")
    out = torch.empty((N, O, H, W), dtype=torch.float32, device=x_nhwc.device)
    with _on(x_nhwc, wpack, bias, out) as ctx:
        _lib.check(lib.svps_conv3x3_pm_small_fwd(_ptr(x_nhwc), _ptr(wpack), _ptr(bias), _ptr(out), N, C, H, W, O, ctx.stream),
                   "svps_conv3x3_pm_small_fwd")
    return out


def semantic_pred(levels, weight, bias):
    """fcn_score = conv1x1(cat(p0, up2(p1), up4(p2), up8(p3))) in one kernel (csrc/semantic_pred.hip): levels = four fp32 NCHW maps
    [N, C, H >> i, W >> i], finest first; weight [K, 4 C, 1, 1] or [K, 4 C]; bias [K] or None -> [N, K, H, W] fp32."""
    lib = _lib.load()
    if len(levels) != 4:
        raise ValueError("semantic_pred takes four pyramid levels")
    N, C, H, W = levels[0].shape
    for i, p in enumerate(levels):
        _need(p, f"levels[{i}]", torch.float32, 4)
        if p.shape != (N, C, H >> i, W >> i):
            raise ValueError(f"levels[{i}] {tuple(p.shape)} != [{N}, {C}, {H >> i}, {W >> i}]")
    w = weight.detach().reshape(weight.shape[0], -1)
    _need(w, "weight", torch.float32, 2)
    K = w.shape[0]
    if w.shape[1] != 4 * C or K > 32 or H % 8 or W % 8:
        raise ValueError("semantic_pred: weight [K <= 32, 4 C] and H, W multiples of 8 expected")
    out = torch.empty((N, K, H, W), dtype=torch.float32, device=levels[0].device)
    with _on(*levels, w, bias, out) as ctx:
        _lib.check(lib.svps_semantic_pred_fwd(_ptr(levels[0]), _ptr(levels[1]), _ptr(levels[2]), _ptr(levels[3]), _ptr(w), _ptr(bias), _ptr(out),
                                              N, C, K, H, W, ctx.stream), "svps_semantic_pred_fwd")
    return out


def group_norm_relu_pm(x, gamma, beta, groups=32, eps=1e-5, want_nchw=False, want_16=None, want_pm=True, stats=None):
    """relu(GroupNorm(groups)(x)) for pixel-major fp32 activations x [N, HW, C] (csrc/gn_relu.hip) -> (y [N, HW, C], y_nchw [N, C, HW]
    or None): the normalisation of the semantic tower in the layout of the deformable-convolution kernel, optionally also in the
    layout the framework's convolutions take. want_16 = torch.bfloat16 / torch.float16: a third result, the same values as 16-bit
    pixel-major rows (what K4 takes as its incoming map); want_16 = "hl": as TWO fp16 planes [2 (hi, lo), N, HW, C] with hi + lo = y to 22
    bits (what K4-HL takes, mode fp16x2: level_fuse_hl_g); want_pm = False drops the fp32 rows (the tower's last layer).
    stats = (partial [N, chunks, 2, C], chunks): per-channel sums from the kernel that produced x (dcn.deform_conv_fused_pm(...,
    gn_stats=True)) - the moments pass over x is skipped."""
    lib = _lib.load()
    _need(x, "x", torch.float32, 3)
    N, HW, C = x.shape
    _need(gamma, "gamma", torch.float32, 1)
    _need(beta, "beta", torch.float32, 1)
    if gamma.numel() != C or beta.numel() != C:
        raise ValueError("group_norm_relu_pm: affine parameters do not match C")
    if want_16 not in (None, torch.bfloat16, torch.float16, "hl"):
        raise ValueError("group_norm_relu_pm: want_16 is torch.bfloat16, torch.float16, 'hl' or None")
    y = torch.empty_like(x) if want_pm else None
    yn = torch.empty((N, C, HW), dtype=torch.float32, device=x.device) if want_nchw else None
    if want_16 == "hl":
        y16 = torch.empty((2, N, HW, C), dtype=torch.float16, device=x.device)
    else:
        y16 = torch.empty((N, HW, C), dtype=want_16, device=x.device) if want_16 is not None else None
    f16_flag = 2 if want_16 == "hl" else int(want_16 == torch.float16)
    ws_bytes = lib.svps_group_norm_relu_workspace_bytes(N, HW, C)
    ws = torch.empty(max(ws_bytes, 4) // 4, dtype=torch.float32, device=x.device)
    with _on(x, gamma, beta, y, yn, y16, ws) as ctx:
        part, chunks = (stats[0], int(stats[1])) if stats is not None else (None, 0)
        if part is not None and (part.dtype != torch.float32 or not part.is_contiguous() or part.numel() != N * chunks * 2 * C):
            raise ValueError("group_norm_relu_pm: stats must be ([N, chunks, 2, C] fp32 contiguous, chunks)")
        rc = lib.svps_group_norm_relu_stats_fwd(_ptr(x), _ptr(part), chunks, _ptr(gamma), _ptr(beta), int(groups), float(eps), _ptr(y),
                                                _ptr(yn), _ptr(y16), f16_flag, _ptr(ws), ws_bytes, N, HW, C, ctx.stream)
    _lib.check(rc, "svps_group_norm_relu_stats_fwd")
    if want_16 is not None:
        return y, yn, y16
    return y, yn


# ---- statistics-fused retriever (csrc/retr_stats.hip, csrc/retr_attn.hip) -------------------------------------------
def retr_stats(feat, H, W, pos_proj, rk, rbk, eps_k, rv, rbv, eps_v):
    """K3': per-pixel reciprocal standard deviations of the key / value LayerNorms, as the aux rows K1' consumes.
    feat [T, H*W, 256] bf16; rk / rv [256, 256] fp16: upper-triangular QR factors of the centred projections; rbk / rbv [256]
    fp32; pos_proj = (Ty [H, 256], Tx [W, 256]) fp32, the position tables already multiplied by the key factor
    (MaskDynamicConv.retr_pos_tables), or None. Returns aux [T, HW, 8] fp16 = one 16-byte row per pixel:
    {1, hi sigma_v, lo sigma_v, 0} fp16, {rstd_k, rstd_v} as raw fp32 (retr_stats_unpack gives the two as fp32 views)."""
    lib = _lib.load()
    mflag = _map_flag(feat)
    T, HW, D = feat.shape
    if HW != H * W:
        raise ValueError("feat rows != H*W")
    _need(rk, "rk", torch.float16, 2)
    _need(rv, "rv", torch.float16, 2)
    if rk.shape != (D, D) or rv.shape != (D, D):
        raise ValueError("rk / rv must be [256, 256]")
    _need(rbk, "rbk", torch.float32, 1)
    _need(rbv, "rbv", torch.float32, 1)
    ytab = xtab = None
    if pos_proj is not None:
        ytab, xtab = pos_proj
        _need(ytab, "ty", torch.float32, 2)
        _need(xtab, "tx", torch.float32, 2)
        if ytab.shape != (H, D) or xtab.shape != (W, D):
            raise ValueError("projected position tables do not match (H, W)")
    aux = torch.empty((T, HW, 8), dtype=torch.float16, device=feat.device)
    with _on(feat, ytab, xtab, rk, rbk, rv, rbv) as ctx:
        rc = lib.svps_retr_stats_fwd(_ptr(feat), _ptr(ytab), _ptr(xtab), _ptr(rk), _ptr(rbk), float(eps_k), _ptr(rv), _ptr(rbv),
                                     float(eps_v), _ptr(aux), T, H, W, D, mflag, ctx.stream)
    _lib.check(rc, "svps_retr_stats_fwd")
    return aux


_ZERO_TABLES = {}


def retr_stats_level(feat, H, W, stages):
    """K3'': the aux rows (see retr_stats) of every retriever stage of one pyramid level from ONE read of the fused map.
    stages: list (1 or 2 entries) of (pos_proj, rk, rbk, eps_k, rv, rbv, eps_v) with the meaning of retr_stats' arguments
    (pos_proj = (Ty [H, 256], Tx [W, 256]) or None). Returns the list of aux tensors [T, HW, 8] fp16."""
    import ctypes
    lib = _lib.load()
    mflag = _map_flag(feat)
    T, HW, D = feat.shape
    if HW != H * W:
        raise ValueError("feat rows != H*W")
    if not 1 <= len(stages) <= 2:
        raise ValueError("a level has one or two retriever stages")
    n = len(stages)
    tys, txs, rks, rbks, rvs, rbvs, auxs, keep = [], [], [], [], [], [], [], []
    for (pos_proj, rk, rbk, eps_k, rv, rbv, eps_v) in stages:
        _need(rk, "rk", torch.float16, 2)
        _need(rv, "rv", torch.float16, 2)
        _need(rbk, "rbk", torch.float32, 1)
        _need(rbv, "rbv", torch.float32, 1)
        if rk.shape != (D, D) or rv.shape != (D, D):
            raise ValueError("rk / rv must be [256, 256]")
        if pos_proj is None:
            key = (H, W, D, feat.device)
            if key not in _ZERO_TABLES:
                _ZERO_TABLES[key] = (torch.zeros((H, D), dtype=torch.float32, device=feat.device),
                                     torch.zeros((W, D), dtype=torch.float32, device=feat.device))
            pos_proj = _ZERO_TABLES[key]
        ty, tx = pos_proj
        _need(ty, "ty", torch.float32, 2)
        _need(tx, "tx", torch.float32, 2)
        if ty.shape != (H, D) or tx.shape != (W, D):
            raise ValueError("projected position tables do not match (H, W)")
        aux = torch.empty((T, HW, 8), dtype=torch.float16, device=feat.device)
        tys.append(ty); txs.append(tx); rks.append(rk); rbks.append(rbk); rvs.append(rv); rbvs.append(rbv); auxs.append(aux)
        keep += [ty, tx, rk, rbk, rv, rbv]
    arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
    fl = lambda i: (ctypes.c_float * n)(*[float(st[i]) for st in stages])
    with _on(feat, *keep) as ctx:
        rc = lib.svps_retr_stats_level_fwd(_ptr(feat), n, arr(tys), arr(txs), arr(rks), arr(rbks), fl(3), arr(rvs), arr(rbvs), fl(6),
                                           arr(auxs), T, H, W, D, mflag, ctx.stream)
    _lib.check(rc, "svps_retr_stats_level_fwd")
    return auxs


def retr_stats_unpack(aux):
    """(rstd_k, rstd_v) [T, HW] fp32: strided views of the two fp32 words of every aux row."""
    w = aux.view(torch.float32)                                  # [T, HW, 4]
    return w[..., 2], w[..., 3]


def retr_slot_pad(L):
    """Rows of the slot axis in the layouts K1' takes: 128 for L <= 128, 256 for L <= 256."""
    return 128 if L <= 128 else 256


def retr_attn(qh, ql, cy, cx, c3, feat, aux, L, H, W, chunks=0):
    """K1': out_ext [T, L, 272] fp32 = {sum_p P rstd_v f_p, sum_p P rstd_v, sum_p P, 0...} with P the softmax over slots of
    rstd_k (Q''.f + cy + cx) + c3 (rstd_k, rstd_v: from the aux rows of retr_stats). qh / ql [T, LP, 256] fp16 (retr_split), cy [T, H, LP], cx [T, W, LP], c3 [T, LP] fp32 with the slot axis
    padded to LP = 128 (L <= 128) or 256 (L <= 256: two passes - probabilities of all slots through the workspace, then P f)."""
    lib = _lib.load()
    _need(qh, "qh", torch.float16, 3)
    _need(ql, "ql", torch.float16, 3)
    mflag = _map_flag(feat)
    _need(aux, "aux", torch.float16, 3)
    for name, x in (("cy", cy), ("cx", cx)):
        _need(x, name, torch.float32, 3)
    _need(c3, "c3", torch.float32, 2)
    T, HW, D = feat.shape
    if not 1 <= L <= 256:
        raise ValueError("the fused retriever covers 1 <= L <= 256 slots")
    LP = retr_slot_pad(L)
    if (HW != H * W or qh.shape != (T, LP, D) or ql.shape != (T, LP, D) or cy.shape != (T, H, LP) or cx.shape != (T, W, LP)
            or c3.shape != (T, LP) or aux.shape != (T, HW, 8)):
        raise ValueError("shape mismatch")
    fwd, name = lib.svps_retr_attn_fwd, "svps_retr_attn_fwd"
    ws_bytes = lib.svps_retr_attn_workspace_bytes(T, L, H, W, chunks)
    ws = torch.empty(max(ws_bytes, 4) // 4, dtype=torch.float32, device=feat.device)
    out = torch.empty((T, L, 272), dtype=torch.float32, device=feat.device)
    with _on(qh, ql, cy, cx, c3, feat, aux) as ctx:
        rc = fwd(_ptr(qh), _ptr(ql), _ptr(cy), _ptr(cx), _ptr(c3), _ptr(feat), _ptr(aux), _ptr(ws), ws_bytes,
                 _ptr(out), T, L, H, W, D, chunks, mflag, ctx.stream)
    _lib.check(rc, name)
    return out


def retr_query_prep(x, lnq_w, lnq_b, lnq_eps, lnk_w, lnk_b, bck, LP):
    """x = to_q(slots) [T, L, 256] fp32 -> (gp [T, LP, 256] = norm_q(x) * lnk_w zero-padded, c3 [T, LP] = log2(e) q . lnk_b
    with -1e30 in the padded rows, a1 [T, LP] = gp . bck)."""
    lib = _lib.load()
    _need(x, "x", torch.float32, 3)
    for name, v in (("lnq_w", lnq_w), ("lnq_b", lnq_b), ("lnk_w", lnk_w), ("lnk_b", lnk_b), ("bck", bck)):
        _need(v, name, torch.float32, 1)
    T, L, D = x.shape
    gp = torch.empty((T, LP, D), dtype=torch.float32, device=x.device)
    c3 = torch.empty((T, LP), dtype=torch.float32, device=x.device)
    a1 = torch.empty((T, LP), dtype=torch.float32, device=x.device)
    with _on(x, lnq_w, lnq_b, lnk_w, lnk_b, bck) as ctx:
        rc = lib.svps_retr_query_prep(_ptr(x), _ptr(lnq_w), _ptr(lnq_b), float(lnq_eps), _ptr(lnk_w), _ptr(lnk_b), _ptr(bck),
                                      _ptr(gp), _ptr(c3), _ptr(a1), T, L, LP, D, ctx.stream)
    _lib.check(rc, "svps_retr_query_prep")
    return gp, c3, a1


def retr_split(q2):
    """fp32 tensor -> (hi, lo) fp16 with hi + lo = q2 to a 22-bit mantissa (the matrix-core operand pair of K1')."""
    lib = _lib.load()
    _need(q2, "q2", torch.float32)
    hi = torch.empty(q2.shape, dtype=torch.float16, device=q2.device)
    lo = torch.empty_like(hi)
    with _on(q2) as ctx:
        _lib.check(lib.svps_retr_split(_ptr(q2), _ptr(hi), _ptr(lo), q2.numel(), ctx.stream), "svps_retr_split")
    return hi, lo


def slot_self_attn(qkv, nheads, split="bf16"):
    """qkv [T, L, 3 * C] fp32 (packed in_proj output: q | k | v, each nheads x 32) -> [T, L, C] = softmax(q k^T / sqrt(32)) v
    per frame and head (nn.MultiheadAttention between its two projections, dynamic_mask_head.py:346-355). split: operand split of the
    matrix products, "bf16" (hi + lo = 16 bits) or "fp16" (22 bits: mode fp16x2)."""
    lib = _lib.load()
    _need(qkv, "qkv", torch.float32, 3)
    T, L, C3 = qkv.shape
    C = C3 // 3
    if C3 != 3 * C or C != nheads * 32:
        raise ValueError("slot_self_attn: head_dim must be 32")
    out = torch.empty((T, L, C), dtype=torch.float32, device=qkv.device)
    with _on(qkv) as ctx:
        fn = lib.svps_slot_self_attn_f16 if split == "fp16" else lib.svps_slot_self_attn
        _lib.check(fn(_ptr(qkv), _ptr(out), T, L, nheads, 32, ctx.stream), "svps_slot_self_attn")
    return out


def pack_b_fragments(weight, split="bf16"):
    """nn.Linear weight [N, K] fp32 -> bf16 [N/32, K/16, 2, 64, 8]: hi / lo halves in MFMA B-fragment order (the layout
    svps_slot_gemm streams: one 1-KiB wave instruction per fragment). N % 32 == 0, K % 16 == 0.
    split="fp16": fp16 hi / lo halves (22 bits of mantissa; svps_slot_gemm_f16), |w| < 65 504."""
    N, K = weight.shape
    if N % 32 or K % 16:
        raise ValueError("pack_b_fragments: N % 32 == 0 and K % 16 == 0 required")
    w = weight.detach().float().contiguous()
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[split]
    if split == "fp16" and float(w.abs().max()) >= 65504.0:
        raise ValueError("pack_b_fragments(split='fp16'): weight outside the fp16 range")
    hi = w.to(dt)
    lo = (w - hi.float()).to(dt)

    def frag(m):                                   # (cb, r, ks, h, j) -> (cb, ks, h, r, j): lane = 32 h + r
        return m.view(N // 32, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).reshape(N // 32, K // 16, 64, 8)
    return torch.stack([frag(hi), frag(lo)], dim=2).contiguous()


ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2


def slot_gemm(x, wpack, bias=None, act=ACT_NONE, out=None):
    """K8: y = act(x @ W^T + bias) for x [..., K] fp32 and wpack = pack_b_fragments(W [N, K]); N % 256 == 0, K % 16 == 0.
    Split-bf16 matrix-core products with fp32 accumulation (fp32-class; see csrc/slot_gemm.hip). A weight packed with split="fp16"
    runs the fp16 hi + lo form (operands within the fp16 range)."""
    lib = _lib.load()
    _need(x, "x", torch.float32)
    f16 = wpack.dtype == torch.float16
    _need(wpack, "wpack", torch.float16 if f16 else torch.bfloat16, 5)
    K = x.shape[-1]
    M = x.numel() // K
    N = wpack.shape[0] * 32
    if wpack.shape[1] * 16 != K or N % 256:
        raise ValueError(f"slot_gemm: x[..., {K}] does not match the packed weight {tuple(wpack.shape)} or N % 256 != 0")
    if bias is not None:
        _need(bias, "bias", torch.float32, 1)
    if out is None:
        out = torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
    elif out.shape != x.shape[:-1] + (N,) or not out.is_contiguous() or out.dtype != torch.float32:
        raise ValueError("slot_gemm: out must be a contiguous fp32 tensor of shape x.shape[:-1] + (N,)")
    with _on(x, wpack, bias, out) as ctx:
        if f16 and act != ACT_NONE:
            _lib.check(lib.svps_slot_gemm_f16_act(_ptr(x), _ptr(wpack), _ptr(bias), _ptr(out), M, K, N, int(act), ctx.stream), "svps_slot_gemm_f16_act")
        elif f16:
            _lib.check(lib.svps_slot_gemm_f16(_ptr(x), _ptr(wpack), _ptr(bias), _ptr(out), M, K, N, ctx.stream), "svps_slot_gemm_f16")
        else:
            _lib.check(lib.svps_slot_gemm(_ptr(x), _ptr(wpack), _ptr(bias), _ptr(out), M, K, N, int(act), ctx.stream), "svps_slot_gemm")
    return out


def slot_gemm_ln(x, wpack, bias, gamma, beta, eps=1e-5, pre=None, post=None, relu=False, out=None):
    """K8 with the LayerNorm step fused into the launch: y = LN(x @ W^T + bias [+ pre]) * gamma + beta (+ReLU) (+post) for
    N = 256; bitwise the result of slot_gemm followed by row_ln. A weight packed with split="fp16": the fp16 hi + lo form."""
    lib = _lib.load()
    _need(x, "x", torch.float32)
    f16 = wpack.dtype == torch.float16
    _need(wpack, "wpack", torch.float16 if f16 else torch.bfloat16, 5)
    K = x.shape[-1]
    M = x.numel() // K
    if wpack.shape[0] * 32 != D_MODEL or wpack.shape[1] * 16 != K:
        raise ValueError(f"slot_gemm_ln: x[..., {K}] does not match the packed weight {tuple(wpack.shape)} or N != 256")
    shape = x.shape[:-1] + (D_MODEL,)
    for name, tns in (("pre", pre), ("post", post)):
        if tns is not None:
            _need(tns, name, torch.float32)
            if tns.numel() != M * D_MODEL:
                raise ValueError(f"{name} shape mismatch")
    _need(gamma, "gamma", torch.float32, 1)
    _need(beta, "beta", torch.float32, 1)
    if bias is not None:
        _need(bias, "bias", torch.float32, 1)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=x.device)
    elif out.shape != shape or not out.is_contiguous() or out.dtype != torch.float32:
        raise ValueError("slot_gemm_ln: out must be a contiguous fp32 tensor of shape x.shape[:-1] + (256,)")
    with _on(x, wpack, bias, pre, post, gamma, beta, out) as ctx:
        fn = lib.svps_slot_gemm_ln_f16 if f16 else lib.svps_slot_gemm_ln
        rc = fn(_ptr(x), _ptr(wpack), _ptr(bias), _ptr(pre), _ptr(post), _ptr(gamma), _ptr(beta), float(eps),
                int(bool(relu)), _ptr(out), M, K, ctx.stream)
    _lib.check(rc, "svps_slot_gemm_ln_f16" if f16 else "svps_slot_gemm_ln")
    return out


def slot_ffn(x, w1pack, b1, w2pack, b2, gamma, beta, eps=1e-5, act=ACT_RELU, pre=None, post=None, out=None):
    """The feed-forward block in one launch (csrc/slot_ffn.hip): y = LN(pre + W2 act(W1 x + b1) + b2) * gamma + beta (+ post);
    x [..., 256] fp32, w1pack = pack_b_fragments(W1 [H, 256]), w2pack = pack_b_fragments(W2 [256, H]), H % 256 == 0.
    Bitwise slot_gemm(act) followed by slot_gemm_ln; the [M, H] hidden tensor is never written."""
    lib = _lib.load()
    _need(x, "x", torch.float32)
    f16 = w1pack.dtype == torch.float16                  # both weights packed with split="fp16": the fp16 hi + lo form
    _need(w1pack, "w1pack", torch.float16 if f16 else torch.bfloat16, 5)
    _need(w2pack, "w2pack", torch.float16 if f16 else torch.bfloat16, 5)
    if x.shape[-1] != D_MODEL:
        raise ValueError("slot_ffn works on rows of 256 values")
    M = x.numel() // D_MODEL
    H = w1pack.shape[0] * 32
    if w1pack.shape[1] * 16 != D_MODEL or w2pack.shape[0] * 32 != D_MODEL or w2pack.shape[1] * 16 != H or H % 256:
        raise ValueError(f"slot_ffn: packed weights {tuple(w1pack.shape)} / {tuple(w2pack.shape)} are not a 256 -> H -> 256 pair with H % 256 == 0")
    if act not in (ACT_RELU, ACT_GELU):
        raise ValueError("slot_ffn: act must be ACT_RELU or ACT_GELU")
    for name, tns in (("pre", pre), ("post", post)):
        if tns is not None:
            _need(tns, name, torch.float32)
            if tns.numel() != M * D_MODEL:
                raise ValueError(f"{name} shape mismatch")
    for name, tns, n in (("b1", b1, H), ("b2", b2, D_MODEL), ("gamma", gamma, D_MODEL), ("beta", beta, D_MODEL)):
        if tns is not None:
            _need(tns, name, torch.float32, 1)
            if tns.numel() != n:
                raise ValueError(f"{name}: {n} values expected")
    if out is None:
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    elif out.shape != x.shape or not out.is_contiguous() or out.dtype != torch.float32:
        raise ValueError("slot_ffn: out must be a contiguous fp32 tensor of x's shape")
    with _on(x, w1pack, b1, w2pack, b2, pre, post, gamma, beta, out) as ctx:
        fn = lib.svps_slot_ffn_f16 if f16 else lib.svps_slot_ffn
        rc = fn(_ptr(x), _ptr(w1pack), _ptr(b1), _ptr(w2pack), _ptr(b2), _ptr(pre), _ptr(post), _ptr(gamma),
                _ptr(beta), float(eps), int(act), _ptr(out), M, H, ctx.stream)
    _lib.check(rc, "svps_slot_ffn_f16" if f16 else "svps_slot_ffn")
    return out


def slot_chain(x, layers):
    """Up to 6 chained 256 -> 256 layers with LayerNorm in ONE launch (csrc/slot_chain.hip). x [..., 256] fp32; layers: list of
    dicts with keys wpack (pack_b_fragments of W [256, 256]), gamma, beta (LayerNorm), and optionally bias, eps (1e-5), relu, pre,
    post ([M, 256] fp32), out (contiguous fp32 [M, 256] or True to allocate; None: the result only feeds the next layer),
    src ("x": the chain's input, "prev": the previous layer's result; default "x" for the first layer, "prev" after).
    Returns the list of results (None where a layer's result was not stored). Bitwise the per-layer slot_gemm_ln launches."""
    lib = _lib.load()
    _need(x, "x", torch.float32)
    if x.shape[-1] != D_MODEL:
        raise ValueError("slot_chain works on rows of 256 values")
    M = x.numel() // D_MODEL
    n = len(layers)
    if not 1 <= n <= 6:
        raise ValueError("slot_chain: 1 .. 6 layers")
    keep, outs = [x], []
    cols = {k: [] for k in ("wpack", "bias", "gamma", "beta", "pre", "post", "out")}
    eps, relu, src = [], [], []
    for i, L in enumerate(layers):
        wp = L["wpack"]
        _need(wp, "wpack", torch.bfloat16, 5)
        if wp.shape[0] * 32 != D_MODEL or wp.shape[1] * 16 != D_MODEL:
            raise ValueError("slot_chain: every layer is 256 -> 256")
        for k in ("gamma", "beta"):
            _need(L[k], k, torch.float32, 1)
        for k in ("bias", "pre", "post"):
            t = L.get(k)
            if t is not None:
                _need(t, k, torch.float32)
                if t.numel() != (D_MODEL if k == "bias" else M * D_MODEL):
                    raise ValueError(f"slot_chain: layer {i} {k} has the wrong size")
        o = L.get("out")
        if o is True or (o is None and i == n - 1):
            o = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        if o is not None and (o.numel() != M * D_MODEL or not o.is_contiguous() or o.dtype != torch.float32):
            raise ValueError("slot_chain: out must be a contiguous fp32 tensor of x's size")
        s_ = L.get("src", "x" if i == 0 else "prev")
        if s_ not in ("x", "prev") or (i == 0 and s_ != "x"):
            raise ValueError("slot_chain: src is 'x' or 'prev' (the first layer reads x)")
        outs.append(o)
        for k, v in (("wpack", wp), ("bias", L.get("bias")), ("gamma", L["gamma"]), ("beta", L["beta"]), ("pre", L.get("pre")),
                     ("post", L.get("post")), ("out", o)):
            cols[k].append(_ptr(v))
            keep.append(v)
        eps.append(float(L.get("eps", 1e-5)))
        relu.append(int(bool(L.get("relu", False))))
        src.append(0 if s_ == "x" else 1)
    vp = lambda v: (ctypes.c_void_p * n)(*v)
    with _on(*[t for t in keep if isinstance(t, torch.Tensor)]) as ctx:
        rc = lib.svps_slot_chain(_ptr(x), M, n, vp(cols["wpack"]), vp(cols["bias"]), vp(cols["gamma"]), vp(cols["beta"]),
                                 (ctypes.c_float * n)(*eps), (ctypes.c_int * n)(*relu), vp(cols["pre"]), vp(cols["post"]),
                                 vp(cols["out"]), (ctypes.c_int * n)(*src), ctx.stream)
    _lib.check(rc, "svps_slot_chain")
    return outs


def bgemm(a, b, bias=None, alpha=1.0, out=None, split="bf16"):
    """K9: C[g, m, n] = alpha * sum_k a[g, m, k] b[g, n, k] (+ bias[g, n]) for fp32 tensors of ANY strides (views, transposes,
    expand()ed batch dimensions): a [G, M, K] or [M, K], b [G, N, K] or [N, K], bias [G, N], [N] or None. Split-bf16 matrix-core
    products with fp32 accumulation (fp32-class; csrc/bgemm.hip). Returns [G, M, N] fp32 (or `out`, any strides).
    split="fp16": operands as fp16 hi + lo (22 bits of mantissa; values within the fp16 range)."""
    lib = _lib.load()
    _need_any(a, "a")
    _need_any(b, "b")
    a3 = a if a.dim() == 3 else a.unsqueeze(0)
    b3 = b if b.dim() == 3 else b.unsqueeze(0)
    if a3.dim() != 3 or b3.dim() != 3 or a3.shape[2] != b3.shape[2]:
        raise ValueError("bgemm: a [G, M, K] and b [G, N, K] must share K")
    G = max(a3.shape[0], b3.shape[0])
    if a3.shape[0] not in (1, G) or b3.shape[0] not in (1, G):
        raise ValueError("bgemm: batch sizes do not broadcast")
    M, K, N = a3.shape[1], a3.shape[2], b3.shape[1]
    sa = [a3.stride(0) if a3.shape[0] == G and G > 1 else 0, a3.stride(1), a3.stride(2)]
    sb = [b3.stride(0) if b3.shape[0] == G and G > 1 else 0, b3.stride(1), b3.stride(2)]
    if out is None:
        out = torch.empty((G, M, N), dtype=torch.float32, device=a.device)
    elif out.shape != (G, M, N) or out.dtype != torch.float32:
        raise ValueError("bgemm: out must be fp32 [G, M, N]")
    sc = list(out.stride())
    sbias = None
    if bias is not None:
        _need_any(bias, "bias")
        b2 = bias if bias.dim() == 2 else bias.unsqueeze(0)
        if b2.shape[-1] != N or b2.shape[0] not in (1, G):
            raise ValueError("bgemm: bias must be [G, N] or [N]")
        sbias = [b2.stride(0) if b2.shape[0] == G and G > 1 else 0, b2.stride(1)]
    arr = lambda v: (ctypes.c_longlong * len(v))(*v)
    with _on(a, b, bias, out) as ctx:
        fn = lib.svps_bgemm_f16 if split == "fp16" else lib.svps_bgemm
        rc = fn(_ptr(a), arr(sa), _ptr(b), arr(sb), _ptr(bias), arr(sbias) if sbias else None, _ptr(out), arr(sc),
                G, M, N, K, float(alpha), ctx.stream)
    _lib.check(rc, "svps_bgemm")
    return out


def _need_any(t, name):
    """fp32 CUDA tensor of any strides (K9 takes element strides)."""
    if not isinstance(t, torch.Tensor) or t.dtype != torch.float32:
        raise TypeError(f"{name}: fp32 tensor expected")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: the HIP path needs a GPU tensor (no CPU fallback)")


# ---- reference precision on the matrix cores: every 16-bit operand as fp16 hi + lo (csrc/level_fuse_hl.hip and the HL forms of
# ---- retr_stats_hl.hip, retr_attn.hip, mask_decode.hip). A level map is ONE tensor [2, T, HW, 256] fp16: plane 0 = hi, plane 1 = lo.
def split_hl(x):
    """fp32 tensor -> [2, *x.shape] fp16 (hi, lo) with hi + lo = x to a 22-bit mantissa (|x| < 65 504)."""
    _need(x, "x", torch.float32)
    hi, lo = retr_split(x.contiguous())
    return torch.stack([hi, lo])


def _need_hl(t, name, T=None):
    _need(t, name, torch.float16, 4)
    if t.shape[0] != 2 or t.shape[3] != D_MODEL or (T is not None and t.shape[1] != T):
        raise ValueError(f"{name}: expected [2 (hi, lo), T, HW, 256] fp16, got {tuple(t.shape)}")


def level_fuse_hl_weights(wc):
    """conv_trans weight [256, 384] fp32 -> the operands of the reference-precision level fusion (csrc/level_fuse_hl.hip):
    wb_hl [2, 256, 128] fp16 planes of W_b = wc[:, 256:] (incoming channels), w0_hl the same for level 0's W_1 + W_2 + W_3 (summed in
    float64), wa_pack = pack_b_fragments(W_a = wc[:, :256], "fp16") for the coarse product g = f W_a^T on K8."""
    w = wc.detach().double()
    wb = w[:, 256:].float().contiguous()
    w0 = (w[:, :128] + w[:, 128:256] + w[:, 256:]).float().contiguous()
    return {"wb_hl": split_hl(wb), "w0_hl": split_hl(w0), "wa_pack": pack_b_fragments(w[:, :256].float().contiguous(), "fp16")}


def level_fuse_hl_composed(wc, bc, n_levels=4, pre=None):
    """Operands of the level recursion WITHOUT any 256-wide product (csrc/level_fuse_hl.hip, round 5): with W = [W_a | W_b] and
    G^(m)_i = f_i (W_a^m)^T, G^(m)_i = up(G^(m+1)_{i-1}) + (W_a^m W_b) x_i + W_a^m b (level 0: W_a^m (W_1 + W_2 + W_3) x_0). Composed in
    float64: {"w": [m] -> [2, 256, 128] fp16 planes of W_a^m W_b, "w0": the same for level 0, "b": [m] -> [256] fp32 = W_a^m b, "b0": the
    level-0 bias}, m < n_levels. pre = (W_t [128, 128(, 1, 1)], b_t [128] or None): a linear 1x1 map x = W_t y + b_t in front of the head
    (the detector's conv_trans, vps_capsule.py:76-79) composed into them, so that the kernel reads y: W_b -> W_b W_t, b -> b + W_b b_t
    (level 0 with W_1 + W_2 + W_3 in the place of W_b)."""
    w = wc.detach().double().reshape(256, 384)
    b = bc.detach().double()
    wa, wb = w[:, :256], w[:, 256:]
    w0 = w[:, :128] + w[:, 128:256] + w[:, 256:]
    b_lvl, b_lvl0 = b, b
    if pre is not None:
        wt = pre[0].detach().double().reshape(pre[0].shape[0], -1)
        if wt.shape != (128, 128):
            raise ValueError(f"level_fuse_hl_composed: pre_linear must map 128 -> 128 channels, got {tuple(wt.shape)}")
        if pre[1] is not None:
            bt = pre[1].detach().double()
            b_lvl, b_lvl0 = b + wb @ bt, b + w0 @ bt
        wb, w0 = wb @ wt, w0 @ wt
    out = {"w": [], "w0": [], "b": [], "b0": []}
    p = torch.eye(256, dtype=torch.float64, device=w.device)
    for m in range(n_levels):
        out["w"].append(split_hl((p @ wb).float().contiguous()))
        out["w0"].append(split_hl((p @ w0).float().contiguous()))
        out["b"].append((p @ b_lvl).float().contiguous())
        out["b0"].append((p @ b_lvl0).float().contiguous())
        p = wa @ p
    return out


def level_fuse_hl_g(cur, gprev, w_hl, bias, H, W, planes=True, f32=False):
    """One launch of K4-HL: out = up(gprev) + w cur + bias. cur [T, 128, H, W] fp32 NCHW, or [2 (hi, lo), T, H*W, 128] fp16 pixel-major
    planes (the semantic tower's own rows: group_norm_relu_pm(want_16="hl")); gprev [T, (H/2)(W/2), 256] fp32 or None; w_hl
    [2, 256, 128] fp16 planes; bias [256] fp32. Returns (planes [2, T, H*W, 256] fp16 or None, fp32 [T, H*W, 256] or None)."""
    lib = _lib.load()
    pm = isinstance(cur, torch.Tensor) and cur.dtype == torch.float16
    if pm:
        _need(cur, "cur", torch.float16, 4)
        T = cur.shape[1]
        if cur.shape != (2, T, H * W, 128):
            raise ValueError(f"cur {tuple(cur.shape)} != [2, T, {H * W}, 128]")
    else:
        _need(cur, "cur", torch.float32, 4)
        T = cur.shape[0]
        if cur.shape != (T, 128, H, W):
            raise ValueError(f"cur {tuple(cur.shape)} != [T, 128, {H}, {W}]")
    _need(bias, "bias", torch.float32, 1)
    _need(w_hl, "w_hl", torch.float16, 3)
    if gprev is not None:
        _need(gprev, "gprev", torch.float32, 3)
        if gprev.shape != (T, (H // 2) * (W // 2), 256) or H % 2 or W % 2:
            raise ValueError(f"gprev {tuple(gprev.shape)} does not match an {H}x{W} level")
    if not (planes or f32):
        raise ValueError("nothing to write")
    out = torch.empty((2, T, H * W, 256), dtype=torch.float16, device=cur.device) if planes else None
    o32 = torch.empty((T, H * W, 256), dtype=torch.float32, device=cur.device) if f32 else None
    with _on(cur, gprev, w_hl, bias) as ctx:
        outs = (_ptr(out[0]) if planes else None, _ptr(out[1]) if planes else None, _ptr(o32), T, H, W, ctx.stream)
        if pm:
            rc = lib.svps_level_fuse_hl_pm_fwd(_ptr(cur[0]), _ptr(cur[1]), _ptr(gprev), _ptr(w_hl[0]), _ptr(w_hl[1]), _ptr(bias), *outs)
        else:
            rc = lib.svps_level_fuse_hl_fwd(_ptr(cur), _ptr(gprev), _ptr(w_hl[0]), _ptr(w_hl[1]), _ptr(bias), *outs)
    _lib.check(rc, "svps_level_fuse_hl_pm_fwd" if pm else "svps_level_fuse_hl_fwd")
    return out, o32


def level_fuse_hl_orders(cur, gprevs, w_hls, biases, H, W):
    """ALL orders of a level in ONE launch of K4-HL (svps_level_fuse_hl_multi_fwd): out_m = up(gprevs[m]) + w_hls[m] cur + biases[m] for
    m = 0 .. n - 1 (n <= 4). cur as in level_fuse_hl_g; gprevs: list of n fp32 [T, (H/2)(W/2), 256] or None (level 0: no upsampled term).
    Returns (planes [2, T, H*W, 256] fp16 of order 0, [None, G^(1), ...] fp32 [T, H*W, 256]); bit-identical to n level_fuse_hl_g calls."""
    import ctypes
    lib = _lib.load()
    n = len(w_hls)
    if not 1 <= n <= 4 or len(biases) != n or (gprevs is not None and len(gprevs) != n):
        raise ValueError("1 ... 4 orders, one weight / bias (/ coarse map) each")
    pm = cur.dtype == torch.float16
    if pm:
        _need(cur, "cur", torch.float16, 4)
        T = cur.shape[1]
        if cur.shape != (2, T, H * W, 128):
            raise ValueError(f"cur {tuple(cur.shape)} != [2, T, {H * W}, 128]")
    else:
        _need(cur, "cur", torch.float32, 4)
        T = cur.shape[0]
        if cur.shape != (T, 128, H, W):
            raise ValueError(f"cur {tuple(cur.shape)} != [T, 128, {H}, {W}]")
    for m in range(n):
        _need(w_hls[m], "w_hl", torch.float16, 3)
        _need(biases[m], "bias", torch.float32, 1)
        if gprevs is not None:
            _need(gprevs[m], "gprev", torch.float32, 3)
            if gprevs[m].shape != (T, (H // 2) * (W // 2), 256) or H % 2 or W % 2:
                raise ValueError(f"gprev {tuple(gprevs[m].shape)} does not match an {H}x{W} level")
    out = torch.empty((2, T, H * W, 256), dtype=torch.float16, device=cur.device)
    o32 = [None] + [torch.empty((T, H * W, 256), dtype=torch.float32, device=cur.device) for _ in range(n - 1)]
    arr = lambda ps: (ctypes.c_void_p * n)(*[p for p in ps])
    with _on(cur, *w_hls, *biases, *(gprevs or [])) as ctx:
        rc = lib.svps_level_fuse_hl_multi_fwd(
            _ptr(cur[0]) if pm else _ptr(cur), _ptr(cur[1]) if pm else None, n,
            arr([g.data_ptr() for g in gprevs]) if gprevs is not None else None,
            arr([w[0].data_ptr() for w in w_hls]), arr([w[1].data_ptr() for w in w_hls]), arr([b.data_ptr() for b in biases]),
            _ptr(out[0]), _ptr(out[1]), arr([0] + [o.data_ptr() for o in o32[1:]]), T, H, W, ctx.stream)
    _lib.check(rc, "svps_level_fuse_hl_multi_fwd")
    return out, o32


def level_fuse_hl(cur, prev_f32, weights, bc, H, W, want_f32=False):
    """K4 at the reference's precision: f = up(prev W_a^T) + W_b cur + b (a 1x1 conv commutes with bilinear interpolation; the 256-wide
    product runs at the coarse resolution on K8 with fp16 hi + lo operands). cur [T, 128, H, W] fp32 NCHW; prev_f32 [T, (H/2)(W/2), 256]
    fp32 (the coarser level's fused map) or None (level 0: cat(x, x, x)); weights = level_fuse_hl_weights(conv weight); bc [256] fp32.
    Returns (planes [2, T, H*W, 256] fp16 (hi, lo), the same values as fp32 [T, H*W, 256] if want_f32 else None).
    dynamic_mask_head.py:171-188."""
    lib = _lib.load()
    _need(cur, "cur", torch.float32, 4)
    T = cur.shape[0]
    if cur.shape != (T, 128, H, W):
        raise ValueError(f"cur {tuple(cur.shape)} != [T, 128, {H}, {W}]")
    _need(bc, "bc", torch.float32, 1)
    g = None
    if prev_f32 is not None:
        _need(prev_f32, "prev_f32", torch.float32, 3)
        if prev_f32.shape != (T, (H // 2) * (W // 2), 256) or H % 2 or W % 2:
            raise ValueError(f"prev_f32 {tuple(prev_f32.shape)} does not match an {H}x{W} level")
        g = slot_gemm(prev_f32.view(-1, 256), weights["wa_pack"])          # [T * HWp, 256] fp32 = f_{i-1} W_a^T
    w_hl = weights["wb_hl"] if prev_f32 is not None else weights["w0_hl"]
    _need(w_hl, "w_hl", torch.float16, 3)
    out = torch.empty((2, T, H * W, 256), dtype=torch.float16, device=cur.device)
    f32 = torch.empty((T, H * W, 256), dtype=torch.float32, device=cur.device) if want_f32 else None
    with _on(cur, g, w_hl, bc) as ctx:
        rc = lib.svps_level_fuse_hl_fwd(_ptr(cur), _ptr(g), _ptr(w_hl[0]), _ptr(w_hl[1]), _ptr(bc), _ptr(out[0]), _ptr(out[1]), _ptr(f32),
                                        T, H, W, ctx.stream)
    _lib.check(rc, "svps_level_fuse_hl_fwd")
    return out, f32


def acc_order_perm(device=None):
    """Column permutation of the fp32 tables svps_retr_stats_hl_fwd takes: table[:, perm] puts factor row 32 B + 8 g + 4 h + j at column
    32 B + 16 h + 4 g + j (a lane's 16 accumulator rows of a row block become 64 contiguous bytes)."""
    b, h, g, j = torch.meshgrid(torch.arange(8), torch.arange(2), torch.arange(4), torch.arange(4), indexing="ij")
    return (32 * b + 8 * g + 4 * h + j).reshape(-1).to(device)


def tile_tx_table(txk):
    """Tx' [W, 256] (accumulator-order columns 32 B + 16 h + 4 g + j), W % 32 == 0 -> the TILED order svps_retr_stats_hl_fwd reads with
    whole-KiB loads: [W / 32][8 B][4 g][2 h][32 pixels][4 j] (flattened back to [W, 256])."""
    W = txk.shape[0]
    return txk.view(W // 32, 32, 8, 2, 4, 4).permute(0, 2, 4, 3, 1, 5).contiguous().view(W, 256)


def retr_stats_hl(feat_hl, H, W, tyk, txk, rk_hi, rk_lo, eps_k, rv_hi, rv_lo, rbv, eps_v, tx_tiled=False):
    """Both LayerNorm statistics of the fused retriever from ONE read of a map given as fp16 hi + lo planes [2, T, HW, 256], both factors
    as fp16 hi + lo (csrc/retr_stats_hl.hip): the aux rows [T, HW, 8] of retr_stats. tyk [H or 1, 256] = Ty + r_k, txk [W or 1, 256] = Tx,
    rbv [256] = r_v: fp32, columns in accumulator order (acc_order_perm)."""
    lib = _lib.load()
    _need_hl(feat_hl, "feat_hl")
    _, T, HW, D = feat_hl.shape
    if HW != H * W:
        raise ValueError("feat rows != H*W")
    for name, m in (("rk_hi", rk_hi), ("rk_lo", rk_lo), ("rv_hi", rv_hi), ("rv_lo", rv_lo)):
        _need(m, name, torch.float16, 2)
        if m.shape != (D, D):
            raise ValueError(f"{name} must be [256, 256]")
    _need(tyk, "tyk", torch.float32, 2)
    _need(txk, "txk", torch.float32, 2)
    _need(rbv, "rbv", torch.float32, 1)
    if tyk.shape[1] != D or txk.shape[1] != D or tyk.shape[0] not in (1, H) or txk.shape[0] not in (1, W) or rbv.numel() != D:
        raise ValueError("tables do not match (H, W)")
    aux = torch.empty((T, HW, 8), dtype=torch.float16, device=feat_hl.device)
    with _on(feat_hl, tyk, txk, rk_hi, rk_lo, rv_hi, rv_lo, rbv) as ctx:
        rc = lib.svps_retr_stats_hl_fwd(_ptr(feat_hl[0]), _ptr(feat_hl[1]), _ptr(tyk), tyk.shape[0], _ptr(txk), txk.shape[0], int(bool(tx_tiled)), _ptr(rk_hi),
                                        _ptr(rk_lo), float(eps_k), _ptr(rv_hi), _ptr(rv_lo), _ptr(rbv), float(eps_v), _ptr(aux), T, H, W, D,
                                        ctx.stream)
    _lib.check(rc, "svps_retr_stats_hl_fwd")
    return aux


def retr_attn_hl(qh, ql, cy, cx, c3, feat_hl, aux, L, H, W, chunks=0):
    """retr_attn with P * rstd_v as fp16 hi + lo on a map given as fp16 hi + lo planes [2, T, HW, 256]; slot axis of qh / ql / cy / cx / c3
    padded to LP = 128 (L <= 128) or 256 (L <= 256: softmax statistics over all slots first, then the retriever once per half of the slots)."""
    lib = _lib.load()
    _need(qh, "qh", torch.float16, 3)
    _need(ql, "ql", torch.float16, 3)
    _need_hl(feat_hl, "feat_hl")
    _need(aux, "aux", torch.float16, 3)
    for name, x in (("cy", cy), ("cx", cx)):
        _need(x, name, torch.float32, 3)
    _need(c3, "c3", torch.float32, 2)
    _, T, HW, D = feat_hl.shape
    if not 1 <= L <= 256:
        raise ValueError("the reference-precision retriever covers 1 <= L <= 256 slots")
    LP = retr_slot_pad(L)
    if (HW != H * W or qh.shape != (T, LP, D) or ql.shape != (T, LP, D) or cy.shape != (T, H, LP) or cx.shape != (T, W, LP)
            or c3.shape != (T, LP) or aux.shape != (T, HW, 8)):
        raise ValueError("shape mismatch")
    ws_bytes = lib.svps_retr_attn_hl_workspace_bytes(T, L, H, W, chunks)
    ws = torch.empty(max(ws_bytes, 4) // 4, dtype=torch.float32, device=feat_hl.device)
    out = torch.empty((T, L, 272), dtype=torch.float32, device=feat_hl.device)
    with _on(qh, ql, cy, cx, c3, feat_hl, aux) as ctx:
        rc = lib.svps_retr_attn_hl_fwd(_ptr(qh), _ptr(ql), _ptr(cy), _ptr(cx), _ptr(c3), _ptr(feat_hl[0]), _ptr(feat_hl[1]), _ptr(aux),
                                       _ptr(ws), ws_bytes, _ptr(out), T, L, H, W, D, chunks, ctx.stream)
    _lib.check(rc, "svps_retr_attn_hl_fwd")
    return out


def mask_decode_hl(feat_hl, embed, bn_scale, bn_shift, fg_scale, fg_shift, want_argmax=False):
    """K2 on a map given as fp16 hi + lo planes [2, T, HW, 256]: mask logits [T, L, HW] fp32 (+ uint8 slot argmax [T, HW])."""
    lib = _lib.load()
    _need_hl(feat_hl, "feat_hl")
    _need(embed, "embed", torch.float32, 3)
    _need(bn_scale, "bn_scale", torch.float32, 1)
    _need(bn_shift, "bn_shift", torch.float32, 1)
    _, T, HW, D = feat_hl.shape
    L = embed.shape[1]
    if embed.shape != (T, L, D) or bn_scale.numel() != D or bn_shift.numel() != D:
        raise ValueError("shape mismatch")
    out = torch.empty((T, L, HW), dtype=torch.float32, device=feat_hl.device)
    amax = torch.empty((T, HW), dtype=torch.uint8, device=feat_hl.device) if want_argmax else None
    with _on(feat_hl, embed, bn_scale, bn_shift) as ctx:
        rc = lib.svps_mask_decode_hl_fwd(_ptr(feat_hl[0]), _ptr(feat_hl[1]), _ptr(embed), _ptr(bn_scale), _ptr(bn_shift), float(fg_scale),
                                         float(fg_shift), _ptr(out), _ptr(amax), T, L, HW, D, ctx.stream)
    _lib.check(rc, "svps_mask_decode_hl_fwd")
    return (out, amax) if want_argmax else out


# ---- exact mode: fp32 storage, fp32 arithmetic (csrc/exact_f32.hip) ---------------------------------------------
F32 = torch.float32


def level_fuse_f32(cur, prev, wT, bc, H, W):
    """Exact-mode K4: cur [T, 128, H, W] fp32, prev [T, (H/2)(W/2), 256] fp32 or None, wT [384, 256] fp32 (the 1x1 conv
    weight transposed), bc [256] -> fused map [T, H*W, 256] fp32 (dynamic_mask_head.py:171-188)."""
    lib = _lib.load()
    _need(cur, "cur", F32, 4)
    _need(wT, "wT", F32, 2)
    _need(bc, "bc", F32, 1)
    T = cur.shape[0]
    if cur.shape != (T, 128, H, W) or wT.shape != (384, 256):
        raise ValueError(f"cur {tuple(cur.shape)} / wT {tuple(wT.shape)}")
    if prev is not None:
        _need(prev, "prev", F32, 3)
        if prev.shape != (T, (H // 2) * (W // 2), 256) or H % 2 or W % 2:
            raise ValueError(f"prev {tuple(prev.shape)} does not match an {H}x{W} level")
    out = torch.empty((T, H * W, 256), dtype=F32, device=cur.device)
    with _on(cur, prev, wT, bc) as ctx:
        _lib.check(lib.svps_level_fuse_f32_fwd(_ptr(cur), _ptr(prev), _ptr(wT), _ptr(bc), _ptr(out), T, H, W, ctx.stream),
                   "svps_level_fuse_f32_fwd")
    return out


def kv_project_f32(feat, H, W, pos_tabs, wkT, bk, lnk_w, lnk_b, lnk_eps, wvT, bv, lnv_w, lnv_b, lnv_eps):
    """Exact-mode K3: feat [T, H*W, 256] fp32, wkT / wvT [256, 256] fp32 TRANSPOSED Linear weights -> k, v fp32."""
    lib = _lib.load()
    _need(feat, "feat", F32, 3)
    T, HW, D = feat.shape
    if HW != H * W:
        raise ValueError("feat rows != H*W")
    for name, x in (("wkT", wkT), ("wvT", wvT)):
        _need(x, name, F32, 2)
    for name, x in (("bk", bk), ("lnk_w", lnk_w), ("lnk_b", lnk_b), ("bv", bv), ("lnv_w", lnv_w), ("lnv_b", lnv_b)):
        _need(x, name, F32, 1)
    ytab = xtab = None
    if pos_tabs is not None:
        ytab, xtab = pos_tabs
        _need(ytab, "pos_y", F32, 2)
        _need(xtab, "pos_x", F32, 2)
        if ytab.shape != (H, D // 2) or xtab.shape != (W, D // 2):
            raise ValueError("pos tables do not match (H, W)")
    k = torch.empty_like(feat)
    v = torch.empty_like(feat)
    with _on(feat, ytab, xtab, wkT, bk, lnk_w, lnk_b, wvT, bv, lnv_w, lnv_b) as ctx:
        rc = lib.svps_kv_project_f32_fwd(_ptr(feat), _ptr(ytab), _ptr(xtab), _ptr(wkT), _ptr(bk), _ptr(lnk_w), _ptr(lnk_b),
                                         float(lnk_eps), _ptr(wvT), _ptr(bv), _ptr(lnv_w), _ptr(lnv_b), float(lnv_eps),
                                         _ptr(k), _ptr(v), T, H, W, D, ctx.stream)
    _lib.check(rc, "svps_kv_project_f32_fwd")
    return k, v


def slot_attn_f32(q, k, v, ln_w, ln_b, eps=1e-5, return_pre_ln=False):
    """Exact-mode K1: q [T, L, 256], k / v [T, HW, 256], all fp32 -> ReLU(LN(softmax_over_slots(q k^T) v)) fp32."""
    lib = _lib.load()
    for name, x in (("q", q), ("k", k), ("v", v)):
        _need(x, name, F32, 3)
    _need(ln_w, "ln_w", F32, 1)
    _need(ln_b, "ln_b", F32, 1)
    T, L, D = q.shape
    HW = k.shape[1]
    if k.shape != (T, HW, D) or v.shape != (T, HW, D):
        raise ValueError("shape mismatch")
    ws_bytes = lib.svps_slot_attn_f32_workspace_bytes(T, L, HW)
    ws = torch.empty(max(ws_bytes, 4) // 4, dtype=F32, device=q.device)
    out = torch.empty((T, L, D), dtype=F32, device=q.device)
    pre = torch.empty_like(out) if return_pre_ln else None
    with _on(q, k, v, ln_w, ln_b) as ctx:
        rc = lib.svps_slot_attn_f32_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(ln_w), _ptr(ln_b), float(eps), _ptr(ws), ws_bytes,
                                        _ptr(out), _ptr(pre), T, L, HW, D, ctx.stream)
    _lib.check(rc, "svps_slot_attn_f32_fwd")
    return (out, pre) if return_pre_ln else out


def mask_decode_f32(feat, embed, bn_scale, bn_shift, fg_scale, fg_shift):
    """Exact-mode K2: feat [T, HW, 256] fp32, embed [T, L, 256] fp32 -> mask logits [T, L, HW] fp32."""
    lib = _lib.load()
    _need(feat, "feat", F32, 3)
    _need(embed, "embed", F32, 3)
    _need(bn_scale, "bn_scale", F32, 1)
    _need(bn_shift, "bn_shift", F32, 1)
    T, HW, D = feat.shape
    L = embed.shape[1]
    if embed.shape != (T, L, D):
        raise ValueError("shape mismatch")
    out = torch.empty((T, L, HW), dtype=F32, device=feat.device)
    with _on(feat, embed, bn_scale, bn_shift) as ctx:
        rc = lib.svps_mask_decode_f32_fwd(_ptr(feat), _ptr(embed), _ptr(bn_scale), _ptr(bn_shift), float(fg_scale),
                                          float(fg_shift), _ptr(out), T, L, HW, D, ctx.stream)
    _lib.check(rc, "svps_mask_decode_f32_fwd")
    return out


class KernelTimer:
    """Device-time accounting of the library's own launches (HIP events on the launch stream): the diagnostics library
    (libslotvps_hip_diag.so) behind the product's launch hook - measurement code, not part of the product path."""

    def __enter__(self):
        lib = _lib.load_diag()
        lib.svps_prof_reset()
        lib.svps_prof_enable(1)
        return self

    def __exit__(self, *exc):
        _lib.load_diag().svps_prof_enable(0)
        return False

    @staticmethod
    def collect(kernel_id):
        lib = _lib.load_diag()
        ms, n = ctypes.c_double(0.0), ctypes.c_int(0)
        _lib.check(lib.svps_prof_collect(kernel_id, ctypes.byref(ms), ctypes.byref(n)), "svps_prof_collect")
        return ms.value, n.value
