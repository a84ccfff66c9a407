"""CPU oracle for the Slot-VPS slot-retriever decode path.  TEST INFRASTRUCTURE ONLY.

This file is a NumPy restatement of the reference's algorithm, written from the formulas in the
reference source (cited per function as file:line relative to the SAITPublic/SlotVPS tree). It is
the checker for the HIP path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import it. Nothing under slotvps_amd/ imports it and the product has no CPU fallback.

Parity status: PINNED.  tests/test_oracle_golden.py checks every function below against golden
vectors that tests/golden/make_golden.py captured by importing the reference's own modules
(mmdet/models/detectors/dynamic_mask_head.py, position_encoding.py) in the build container and
running them on seeded inputs (fixtures under tests/golden/). The reference has no tests or
known-answer vectors of its own, so these captured outputs are the pin.
generate_final_outputs (vps_temporal_slots.py:144-160) is pinned by executing the torch ops of
those lines on the same seeded inputs inside tests/golden/make_golden.py (the enclosing module needs
mmcv, which the container lacks). Exception: deform_conv (f2-ii) is parity-UNPINNED, see its docstring.

Parameters are passed as flat dicts keyed by the reference's state_dict names, e.g.
``head_series_2.1.inst_interact.to_k.weight``, so captured checkpoints plug in unchanged.

Arithmetic type: every function takes ``dt`` (np.float32 = the reference's arithmetic, np.float64
for a tighter yardstick).  ``Storage`` models the HIP path's storage policy (which tensors are
rounded to bf16 when they are written to HBM); ``Storage.exact()`` rounds nothing.
"""
import math

import numpy as np
from scipy.special import erf as _erf

LN_EPS = 1e-5
BN_EPS = 1e-5


# --------------------------------------------------------------------------------------------
# storage policy
# --------------------------------------------------------------------------------------------
def round_bf16(x):
    """Round-to-nearest-even to bfloat16, returned as float32 (NaN/Inf preserved)."""
    a = np.ascontiguousarray(x, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    rounded = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    out = rounded.astype(np.uint32).view(np.float32)
    special = ~np.isfinite(a)
    if special.any():
        out = np.where(special, a, out)
    return out.reshape(a.shape)


class Storage:
    """Which tensors the path keeps in bf16 in HBM. Each hook returns float32/float64 values.

    bf16=False rounds nothing (the reference's fp32 arithmetic). With bf16=True the rounding points
    are: the fused level map, the operands of the 1x1 fusion conv and of the k/v projections
    (activations and weight matrices), and the post-LayerNorm q / k / v. Biases, accumulation,
    LayerNorm statistics and everything on the slot side stay fp32. ``torch_conv`` additionally models
    the interim PyTorch level-fusion conv of the HIP path (a bf16 conv also rounds its bias)."""

    def __init__(self, bf16, torch_conv=False, kv_bf16=True, fmt="bf16"):
        self.bf16 = bool(bf16)
        self.torch_conv = bool(torch_conv)
        self.kv_bf16 = bool(kv_bf16)
        self.fmt = fmt                 # the 16-bit format of the rounding points: "bf16" or "fp16" (same bytes, 3 more mantissa bits)

    @classmethod
    def exact(cls):
        return cls(False)

    @classmethod
    def bf16_policy(cls, torch_conv=False):
        """The first form of the fast path (K3 + K1): bf16 fused map AND bf16 projection operands / weights / q / k / v."""
        return cls(True, torch_conv)

    @classmethod
    def fused_policy(cls):
        """The statistics-fused retriever (K3' + K1'): only the fused level map (and the 1x1 fusion conv's operands) are
        bf16; nothing on the projection / q / k / v side is rounded as a tensor."""
        return cls(True, False, kv_bf16=False)

    @classmethod
    def fused_fp16_policy(cls):
        """fused_policy with fp16 instead of bf16 level maps and level-fusion operands (MultiScaleDynamicMaskHead.map_dtype = "fp16")."""
        return cls(True, False, kv_bf16=False, fmt="fp16")

    def _round(self, x):
        if self.fmt == "fp16":
            return np.asarray(x).astype(np.float16).astype(x.dtype)
        return round_bf16(x).astype(x.dtype)

    def _r(self, x):
        return self._round(x) if self.bf16 else x

    def _rt(self, x):
        return self._round(x) if (self.bf16 and self.torch_conv) else x

    def _rk(self, x):
        return self._round(x) if (self.bf16 and self.kv_bf16) else x

    feat = _r        # fused level feature map f (input of the projections, the decode, the next level)
    conv_in = _r     # concat(upsampled previous level, current 128-ch map): operand of the 1x1 conv
    proj_in = _rk    # f + pos, the k-projection operand
    weight = _r      # level-fusion conv weight matrix as a matrix-core operand
    proj_weight = _rk  # to_k / to_v weight matrices as matrix-core operands
    kv = _rk         # post-LayerNorm k and v
    q = _rk          # post-LayerNorm q
    conv_bias = _rt  # bias of the level-fusion conv (bf16 only in the interim torch conv)


# --------------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------------
def layer_norm(x, w, b, eps=LN_EPS):
    """torch.nn.LayerNorm over the last dim (biased variance)."""
    mu = x.mean(axis=-1, keepdims=True)
    xc = x - mu
    var = (xc * xc).mean(axis=-1, keepdims=True)
    return xc / np.sqrt(var + x.dtype.type(eps)) * w + b


def linear(x, w, b=None):
    y = x @ w.T
    return y if b is None else y + b


def relu(x):
    return np.maximum(x, 0)


def gelu(x):
    """F.gelu default (exact erf form)."""
    return (0.5 * x * (1.0 + _erf(x / math.sqrt(2.0)))).astype(x.dtype)


def softmax(x, axis):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


def _act(name):
    if name == "relu":
        return relu
    if name == "gelu":
        return gelu
    raise RuntimeError(f"activation should be relu/gelu, not {name}.")


def _p(params, prefix, name, dt):
    return np.asarray(params[prefix + name], dtype=dt)


# --------------------------------------------------------------------------------------------
# a7  sine position embedding          position_encoding.py:236-256 (normalize=True, scale=2*pi)
# --------------------------------------------------------------------------------------------
def pos_embed_sine(H, W, D=256, temperature=10000.0):
    """Returns [H*W, D] float32, pixel-major (row = h*W + w): the 'b h w c' view of the reference's
    [1, D, H, W] output. Every operation in float32, in the reference's order."""
    f32 = np.float32
    npf = D // 2
    y = np.arange(1, H + 1, dtype=f32)                      # cumsum of ones          :241
    x = np.arange(1, W + 1, dtype=f32)                      #                         :242
    eps, scale = f32(1e-6), f32(2 * math.pi)
    y = y / (y[-1] + eps) * scale                           # :245
    x = x / (x[-1] + eps) * scale                           # :246
    i = np.arange(npf, dtype=f32)
    dim_t = np.power(f32(temperature), (f32(2) * np.floor(i / f32(2)) / f32(npf)).astype(f32)).astype(f32)  # :248-249
    py = y[:, None] / dim_t                                 # [H, npf]                :252
    px = x[:, None] / dim_t                                 # [W, npf]                :251
    ey = np.where((np.arange(npf) % 2) == 0, np.sin(py), np.cos(py)).astype(f32)   # interleave sin/cos :253-254
    ex = np.where((np.arange(npf) % 2) == 0, np.sin(px), np.cos(px)).astype(f32)
    out = np.empty((H, W, D), dtype=f32)
    out[:, :, :npf] = ey[:, None, :]                        # cat((pos_y, pos_x))     :255
    out[:, :, npf:] = ex[None, :, :]
    return out.reshape(H * W, D)


# --------------------------------------------------------------------------------------------
# a1  slot <-> pixel retriever          dynamic_mask_head.py:423-461  (MaskDynamicConv.forward)
# --------------------------------------------------------------------------------------------
def retriever_core(q, k, v, ln_w, ln_b, eps=LN_EPS, return_pre=False):
    """K1 boundary. q [L, D], k/v [HW, D] -> ReLU(LN(softmax_over_slots(q k^T) v)) [L, D].
    :435 logits unscaled, :446 softmax over dim=1 (the slot axis), :456 plain sum over pixels,
    :458-459 norm1 + ReLU."""
    logits = q @ k.T                                  # [L, HW]
    p = softmax(logits, axis=0)                       # each pixel column sums to 1 over slots
    pre = p @ v                                       # [L, D]
    out = relu(layer_norm(pre, ln_w, ln_b, eps))
    return (out, pre) if return_pre else out


def retriever_project(slots, feat, pos, params, prefix, st, dt):
    """The three projections + LayerNorms of :431-433 under a storage policy.
    slots [L, D], feat [HW, D] pixel-major, pos [HW, D] or None."""
    g = lambda n: _p(params, prefix, n, dt)
    q = layer_norm(linear(slots, g("to_q.weight"), g("to_q.bias")), g("norm_q.weight"), g("norm_q.bias"))
    kin = st.proj_in(feat + pos) if pos is not None else feat
    k = layer_norm(linear(kin, st.proj_weight(g("to_k.weight")), g("to_k.bias")), g("norm_k.weight"), g("norm_k.bias"))
    v = layer_norm(linear(feat, st.proj_weight(g("to_v.weight")), g("to_v.bias")), g("norm_v.weight"), g("norm_v.bias"))
    return st.q(q), st.kv(k), st.kv(v)


def retriever(slots, feat, pos, params, prefix, st=None, dt=np.float32):
    """MaskDynamicConv.forward: slots [L, D], feat/pos [HW, D] (pixel-major) -> [L, D]."""
    st = st or Storage.exact()
    q, k, v = retriever_project(slots.astype(dt), feat.astype(dt), None if pos is None else pos.astype(dt),
                                params, prefix, st, dt)
    return retriever_core(q, k, v, _p(params, prefix, "norm1.weight", dt), _p(params, prefix, "norm1.bias", dt))


# --------------------------------------------------------------------------------------------
# a2  slot <-> slot retriever + temporal head   dynamic_mask_head.py:550-572, :494-527
# --------------------------------------------------------------------------------------------
def slots_retriever(cur, prev, params, prefix, dt=np.float32):
    """SlotsDynamicConv.forward: cur [Lq, D] queries, prev [Lk, D] keys/values (no pos, :556);
    softmax over the QUERY axis (dim=1 of [1, Lq, Lk], :562)."""
    g = lambda n: _p(params, prefix, n, dt)
    q = layer_norm(linear(cur, g("to_q.weight"), g("to_q.bias")), g("norm_q.weight"), g("norm_q.bias"))
    k = layer_norm(linear(prev, g("to_k.weight"), g("to_k.bias")), g("norm_k.weight"), g("norm_k.bias"))
    v = layer_norm(linear(prev, g("to_v.weight"), g("to_v.bias")), g("norm_v.weight"), g("norm_v.bias"))
    attn = softmax(q @ k.T, axis=0)
    return relu(layer_norm(attn @ v, g("norm1.weight"), g("norm1.bias")))


def temporal_head(S, params, prefix, activation="relu", dt=np.float32):
    """TemporalSlotsHead.forward with features == mask_query == S [T*L, D] (:313-315 call site).
    x = norm2(S + retriever); x = norm3(x + FFN(x)). norm1 of this module is unused."""
    g = lambda n: _p(params, prefix, n, dt)
    x = S + slots_retriever(S, S, params, prefix + "inst_interact.", dt)          # :511-515
    x = layer_norm(x, g("norm2.weight"), g("norm2.bias"))                          # :517
    y = linear(_act(activation)(linear(x, g("linear1.weight"), g("linear1.bias"))),
               g("linear2.weight"), g("linear2.bias"))                             # :520
    return layer_norm(x + y, g("norm3.weight"), g("norm3.bias"))                   # :524-525


# --------------------------------------------------------------------------------------------
# a3 / a4 / a5  one stage               dynamic_mask_head.py:291-400  (MaskRCNNHead)
# --------------------------------------------------------------------------------------------
def multihead_self_attention(x, params, prefix, nhead, dt=np.float32):
    """nn.MultiheadAttention(D, nhead)(x, x, x) for one sequence x [L, D] (batch 1, :352)."""
    g = lambda n: _p(params, prefix, n, dt)
    L, D = x.shape
    hd = D // nhead
    qkv = linear(x, g("in_proj_weight"), g("in_proj_bias"))
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    q = q * dt(math.sqrt(1.0 / hd))
    qh = q.reshape(L, nhead, hd).transpose(1, 0, 2)
    kh = k.reshape(L, nhead, hd).transpose(1, 0, 2)
    vh = v.reshape(L, nhead, hd).transpose(1, 0, 2)
    attn = softmax(qh @ kh.transpose(0, 2, 1), axis=-1)
    o = (attn @ vh).transpose(1, 0, 2).reshape(L, D)
    return linear(o, g("out_proj.weight"), g("out_proj.bias"))


def stage_till_ffn(slots, feat, pos, params, prefix, nhead=8, activation="gelu", st=None, dt=np.float32,
                   retriever_fn=None):
    """forward_till_ffn :342-388. slots [L, D]; feat/pos [HW, D]. retriever_fn lets a test splice the
    HIP kernel's result in at the K1 boundary."""
    g = lambda n: _p(params, prefix, n, dt)
    s = slots.astype(dt)
    s1 = layer_norm(s + multihead_self_attention(s, params, prefix + "self_attn.", nhead, dt),
                    g("norm1.weight"), g("norm1.bias"))                             # :352-358
    fn = retriever_fn or retriever
    r = fn(s1, feat, pos, params, prefix + "inst_interact.", st, dt)                 # :368
    s2 = layer_norm(s1 + r.astype(dt), g("norm2.weight"), g("norm2.bias"))           # :374-376
    y = linear(_act(activation)(linear(s2, g("linear1.weight"), g("linear1.bias"))),
               g("linear2.weight"), g("linear2.bias"))                              # :379
    return layer_norm(s2 + y, g("norm3.weight"), g("norm3.bias"))                    # :384-385


def stage_after_ffn(obj, params, prefix, num_cls=2, num_reg=2, dt=np.float32):
    """forward_after_ffn :390-400 -> (class_logits [L, C], slot embedding [L, D])."""
    g = lambda n: _p(params, prefix, n, dt)
    c = obj
    for i in range(num_cls):
        c = relu(layer_norm(linear(c, g(f"cls_module.{3 * i}.weight")),
                            g(f"cls_module.{3 * i + 1}.weight"), g(f"cls_module.{3 * i + 1}.bias")))
    e = obj
    for i in range(num_reg):
        e = relu(layer_norm(linear(e, g(f"reg_module.{3 * i}.weight")),
                            g(f"reg_module.{3 * i + 1}.weight"), g(f"reg_module.{3 * i + 1}.bias")))
    logits = linear(c, g("class_logits.weight"), g("class_logits.bias"))
    return logits, e


def stage(slots_t, feat_t, pos_t, params, prefix, temporal, cfg, st=None, dt=np.float32, retriever_fn=None):
    """MaskRCNNHead.forward :291-340 for T frames. Returns (logits[T], embeds[T])."""
    T = len(slots_t)
    objs = [stage_till_ffn(slots_t[t], feat_t[t], pos_t[t], params, prefix, cfg["nhead"], cfg["activation"],
                           st, dt, retriever_fn) for t in range(T)]
    if temporal:
        S = np.concatenate(objs, axis=0)                                            # :310
        S = S + temporal_head(S, params, prefix + "temporal_query_head.", cfg["temporal_activation"], dt)  # :313-317
        objs = np.split(S, T, axis=0)                                               # :320-322
    outs = [stage_after_ffn(o, params, prefix, cfg["num_cls"], cfg["num_reg"], dt) for o in objs]
    return [o[0] for o in outs], [o[1] for o in outs]


# --------------------------------------------------------------------------------------------
# a6  multi-scale head                  dynamic_mask_head.py:138-228
# --------------------------------------------------------------------------------------------
def upsample2x_bilinear(x):
    """F.interpolate(x, None, 2, mode='bilinear', align_corners=False) for x [C, H, W] (:178)."""
    C, H, W = x.shape

    def taps(n):
        dst = np.arange(2 * n, dtype=np.float64)
        src = np.maximum((dst + 0.5) * 0.5 - 0.5, 0.0)
        i0 = np.floor(src).astype(np.int64)
        i1 = np.minimum(i0 + 1, n - 1)
        lam = (src - i0).astype(x.dtype)
        return i0, i1, lam

    y0, y1, ly = taps(H)
    x0, x1, lx = taps(W)
    rows = x[:, y0, :] * (1 - ly)[None, :, None] + x[:, y1, :] * ly[None, :, None]
    return rows[:, :, x0] * (1 - lx)[None, None, :] + rows[:, :, x1] * lx[None, None, :]


def fuse_level(x, prev, wc, bc, st=None):
    """One level of the feature-side glue (:171-188). x [128, H, W] incoming map, prev [256, H/2, W/2]
    fused map of the coarser level or None (level 0: cat(x, x, x), :183), wc [256, 384], bc [256]
    -> fused map, pixel-major [H*W, 256]."""
    st = st or Storage.exact()
    C, H, W = x.shape
    if prev is None:
        cat = np.concatenate([x, x, x], axis=0)                               # :183
    else:
        cat = np.concatenate([upsample2x_bilinear(prev), x], axis=0)          # :178-179
    cat = st.conv_in(cat)
    y = st.weight(wc) @ cat.reshape(cat.shape[0], H * W) + st.conv_bias(bc)[:, None]   # 1x1 conv :181/:185
    return st.feat(np.ascontiguousarray(y.T))


DEFAULT_CFG = dict(nhead=8, activation="gelu", temporal_activation="relu", num_cls=2, num_reg=2,
                   per_level_stages=(1, 2, 2, 2), temporal_stages=(3, 4, 5, 6))


def head_forward(features, init_slots, pos, params, cfg=None, st=None, dt=np.float32, retriever_fn=None,
                 fused_override=None):
    """MultiScaleDynamicMaskHead.forward :138-228, merge_operation='concat', trans_in_dim=384.

    features[t][i]: [128, Hi, Wi] (coarse -> fine), init_slots [L, D] (same embedding for all frames,
    vps_temporal_slots.py:286), pos[i]: [Hi*Wi, D] pixel-major (identical for every frame).
    Returns (logits [T][n_stage][L, C], embeds [T][n_stage][L, D], fused [T][4] each [Hi*Wi, D] pixel-major).
    fused_override[t][i] ([Hi*Wi, D]) replaces the computed fused map of level i (teacher forcing: lets a
    test compare everything downstream of the level fusion on identical feature maps).
    """
    cfg = dict(DEFAULT_CFG, **(cfg or {}))
    st = st or Storage.exact()
    T, nlev = len(features), len(features[0])
    wc = _p(params, "", "conv_trans.conv.weight", dt).reshape(256, -1)      # [256, 384]
    bc = _p(params, "", "conv_trans.conv.bias", dt)
    slots = [np.asarray(init_slots, dtype=dt) for _ in range(T)]
    logits = [[] for _ in range(T)]
    embeds = [[] for _ in range(T)]
    fused = [[None] * nlev for _ in range(T)]
    prev = [None] * T                                                        # [256, H, W] per frame
    stage_idx = 0
    for i in range(nlev):
        cur_pm = []
        for t in range(T):
            x = np.asarray(features[t][i], dtype=dt)
            y = fuse_level(x, prev[t], wc, bc, st)                            # :171-188
            if fused_override is not None:
                y = np.asarray(fused_override[t][i], dtype=dt)
            cur_pm.append(y)
            prev[t] = np.ascontiguousarray(y.T).reshape(256, x.shape[1], x.shape[2])
            fused[t][i] = y
        for j in range(cfg["per_level_stages"][i]):
            prefix = f"head_series_{i}.{j}."
            lg, em = stage(slots, cur_pm, [pos[i]] * T, params, prefix, stage_idx in cfg["temporal_stages"],
                           cfg, st, dt, retriever_fn)
            for t in range(T):
                logits[t].append(lg[t])
                embeds[t].append(em[t])
            slots = em                                                        # :210-211
            stage_idx += 1
    return logits, embeds, fused


# --------------------------------------------------------------------------------------------
# a8  slot -> mask decode               vps_temporal_slots.py:144-160
# --------------------------------------------------------------------------------------------
def bn_eval_affine(weight, bias, mean, var, eps=BN_EPS):
    """Fold an eval-mode BatchNorm into (scale, shift)."""
    scale = weight / np.sqrt(var + eps)
    return scale, bias - mean * scale


def mask_decode(feat, embed, bn_scale, bn_shift, fg_scale, fg_shift):
    """feat [HW, D] pixel-major finest fused map, embed [L, D] -> mask logits [L, HW].
    :146 feat_bn (eval), :147 F.normalize(p=2, dim=1, eps=1e-12), :149 einsum, :153 fg_bn (scalar affine)."""
    g = feat * bn_scale + bn_shift
    nrm = np.maximum(np.sqrt((g * g).sum(axis=1, keepdims=True)), 1e-12)
    gh = g / nrm
    m = embed @ gh.T
    return m * fg_scale + fg_shift


def slot_argmax(mask_logits):
    """argmax over the slot axis per pixel (first maximum wins), uint8. mask_logits [L, HW]."""
    return np.argmax(mask_logits, axis=0).astype(np.uint8)


# --------------------------------------------------------------------------------------------
# f2-ii  deformable convolution (DCNv1)   mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu:82-114, :190-241
# --------------------------------------------------------------------------------------------
def deform_conv(x, offset, weight, stride=1, padding=0, dilation=1, deformable_groups=1):
    """x [C, H, W], offset [dg*2*kh*kw, Ho, Wo] (channel 2*(i*kw+j) = dy, +1 = dx), weight [O, C, kh, kw]
    -> [O, Ho, Wo]. Published DCNv1 forward as the reference's CUDA kernel evaluates it: sample at
    (h_in + i*dil + dy, w_in + j*dil + dx), zero outside (-1, H) x (-1, W), bilinear with out-of-image corners = 0,
    then the ordinary convolution sum (im2col + addmm, deform_conv_cuda.cpp:152-258).
    PARITY UNPINNED for this function: the reference's op is CUDA-only (deform_conv.py:44-45) and cannot run in the build
    container, and the reference holds no vectors for it; what the tests check are properties of the published DCNv1
    formula (zero offsets == conv2d, integer offsets == shifted conv, fractional offsets == the hand-computed bilinear
    blend, and nine hand-worked samples on and around the image border for the rules of deform_conv_cuda_kernel.cu:82-114 /
    :224 - tests/test_deform_conv.py). The semantic-tower fixture that uses it is therefore not an independent pin either."""
    x = np.asarray(x)
    C, H, W = x.shape
    O, _, kh, kw = weight.shape
    Ho = (H + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
    cpg = C // deformable_groups
    ys, xs = np.meshgrid(np.arange(Ho), np.arange(Wo), indexing="ij")
    cols = np.zeros((C, kh * kw, Ho, Wo), dtype=x.dtype)
    for g in range(deformable_groups):
        for i in range(kh):
            for j in range(kw):
                t = i * kw + j
                hf = (ys * stride - padding + i * dilation).astype(x.dtype) + offset[g * 2 * kh * kw + 2 * t]
                wf = (xs * stride - padding + j * dilation).astype(x.dtype) + offset[g * 2 * kh * kw + 2 * t + 1]
                inside = (hf > -1) & (wf > -1) & (hf < H) & (wf < W)
                hl, wl = np.floor(hf).astype(np.int64), np.floor(wf).astype(np.int64)
                hh, wh = hl + 1, wl + 1
                lh, lw = hf - hl, wf - wl
                uh, uw = 1 - lh, 1 - lw

                def tap(hi, wi, ok):
                    ok = ok & inside
                    v = x[g * cpg:(g + 1) * cpg][:, np.clip(hi, 0, H - 1), np.clip(wi, 0, W - 1)]
                    return np.where(ok[None], v, 0)
                v1 = tap(hl, wl, (hl >= 0) & (wl >= 0))
                v2 = tap(hl, wh, (hl >= 0) & (wh <= W - 1))
                v3 = tap(hh, wl, (hh <= H - 1) & (wl >= 0))
                v4 = tap(hh, wh, (hh <= H - 1) & (wh <= W - 1))
                cols[g * cpg:(g + 1) * cpg, t] = (uh * uw) * v1 + (uh * lw) * v2 + (lh * uw) * v3 + (lh * lw) * v4
    return np.einsum("ock,ckhw->ohw", weight.reshape(O, C, kh * kw), cols)
