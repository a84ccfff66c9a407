"""Host-side mirror of the reference's slot head (mmdet/models/detectors/dynamic_mask_head.py):
same class names, constructor arguments, parameter names (checkpoint keys) and call signature, but
the whole clip is processed as one batch and the slot<->pixel retriever runs on the HIP kernel K1.

What runs where
  * K1 (HIP, libslotvps_hip.so): logits, softmax over slots, attn.v, LayerNorm, ReLU for all T frames
    of a stage in one launch (MaskDynamicConv.forward lines 435-459).
  * K3 (HIP): the pixel-side projections k = norm_k(to_k(f + pos)), v = norm_v(to_v(f)) (:432-433)
    for all T frames, written as the bf16 tensors K1 streams.
  * K4 (HIP): the level fusion  f_i = conv1x1(cat(bilinear_x2(f_{i-1}), x_i))  (:171-188).
  * K5 (HIP): every  residual + LayerNorm (+ReLU) (+cast)  step of the slot update, one launch each.
  * PyTorch-ROCm (plumbing around the kernels, same math as the reference lines cited inline): the small
    GEMMs of the slot side on [T, L, 256] tensors (self-attention, FFN, temporal head, towers).
There is no CPU path: modules raise if their tensors are not on a GPU.

Storage policy (what is rounded to bf16 in HBM): the fused level map f, the projection operand f+pos,
the projection weights, and the post-LayerNorm q / k / v. Slot-side tensors stay fp32.

Exact mode (`head.set_mode("fp32")`): the reference runs this path in fp32 (vps_temporal_slots.py:55); in this
mode nothing is stored below fp32 and the pixel side runs on the fp32 kernels of csrc/exact_f32.hip (level fusion,
projections + LayerNorm, retriever, decode), the slot-side self-attention on explicit fp32 matrix products. One to two
orders of magnitude slower; it exists so that the head can be compared free-running with the reference's own outputs.
"""
import copy
import math

import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .registry import HEADS

BF16 = torch.bfloat16
# The mode every module of this file starts in (MultiScaleDynamicMaskHead.MODES): "fp16x2" is the fastest mode that meets the north star's
# tolerance against the reference's own fp32 outputs at full size (1e-4 on the mask logits; tests/test_full_size_gpu.py). The 16-bit
# storage policies ("bf16", "fp16") are opt-in: other_config=dict(mode="bf16") / head.set_mode("bf16") - 2x the rate, 1.4e-2 / 1.9e-3 from
# the reference.
DEFAULT_MODE = "fp16x2"
DEFAULT_PRECISION = "fp16x2"


def _get_activation_fn(activation):
    if activation == "relu":
        return F.relu
    if activation == "gelu":
        return F.gelu
    if activation == "glu":
        return F.glu
    raise RuntimeError(f"activation should be relu/gelu, not {activation}.")


def _get_clones(module, n):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(n)])


def _cached(module, name, params, build):
    """Derived tensors (stacked / transposed / bf16 copies of parameters), rebuilt when a source changes."""
    key = tuple((p.data_ptr(), p._version) for p in params)
    store = module.__dict__.setdefault("_derived", {})
    hit = store.get(name)
    if hit is None or hit[0] != key:
        with torch.no_grad():
            store[name] = (key, build())
    return store[name][1]


def fast_linear(owner, name, x, weight, bias=None, act=None, out=None, split="bf16"):
    """y = act(x @ weight^T + bias) for the dense layers of the slot update. bf16 mode: K8 (csrc/slot_gemm.hip: matrix cores,
    split-bf16 products, fp32 accumulation - fp32-class) with the weight packed once per version into fragment order;
    exact mode (owner.precision == "fp32") or shapes K8 does not cover: the GEMM library in fp32.
    act: None, "relu" or "gelu"."""
    N, K = weight.shape
    prec = getattr(owner, "precision", "bf16")
    if prec == "fp32" or N % 256 or K % 16 or not x.is_cuda or not owner.use_slot_gemm:
        y = F.linear(x, weight, bias)
        y = F.relu(y) if act == "relu" else (F.gelu(y) if act == "gelu" else y)
        if out is not None:
            out.copy_(y)
            return out
        return y
    if prec == "fp16x2":
        # reference precision on the matrix cores: both operands as fp16 hi + lo (22 bits; K8's F16 form, 2.8e-6 against float64 where the
        # bf16 split measures 2.5e-5 and the library's fp32 4e-6 ... 1e-5), the activation in the kernel's epilogue
        wp = _cached(owner, "wp_" + name + "_f16", [weight], lambda: ops.pack_b_fragments(weight, "fp16"))
        code = {None: ops.ACT_NONE, "relu": ops.ACT_RELU, "gelu": ops.ACT_GELU}[act]
        return ops.slot_gemm(x.contiguous(), wp, bias, code, out)
    # split="fp16": operands as fp16 hi + lo (22 bits) instead of bf16 hi + lo (16): the query side of the fused retriever
    wp = _cached(owner, "wp_" + name + ("_f16" if split == "fp16" else ""), [weight], lambda: ops.pack_b_fragments(weight, split))
    code = {None: ops.ACT_NONE, "relu": ops.ACT_RELU, "gelu": ops.ACT_GELU}[act]
    return ops.slot_gemm(x.contiguous(), wp, bias, code, out)


def fast_linear_ln(owner, name, x, weight, bias, norm, pre=None, post=None, relu=False, out=None):
    """LN(x @ weight^T + bias [+ pre]) (+ReLU) (+post) for a 256-column layer followed by `norm` (nn.LayerNorm or a
    (gamma, beta, eps) triple): bf16 mode = ONE launch of K8 with the LayerNorm epilogue (csrc/slot_gemm.hip), bitwise the
    two-launch form; otherwise fast_linear + K5."""
    gamma, beta, eps = (norm.weight, norm.bias, norm.eps) if isinstance(norm, nn.LayerNorm) else norm
    N, K = weight.shape
    prec = getattr(owner, "precision", "bf16")
    if prec == "fp16x2" and N == 256 and K % 16 == 0 and x.is_cuda and owner.use_slot_gemm and getattr(owner, "fuse_ln", True):
        wp = _cached(owner, "wp_" + name + "_f16", [weight], lambda: ops.pack_b_fragments(weight, "fp16"))   # K8's fp16-split form + LN epilogue
        return ops.slot_gemm_ln(x.contiguous(), wp, bias, gamma, beta, eps, pre=None if pre is None else pre.contiguous(),
                                post=None if post is None else post.contiguous(), relu=relu, out=out)
    if (prec in ("fp32", "fp16x2") or N != 256 or K % 16 or not x.is_cuda or not owner.use_slot_gemm
            or not getattr(owner, "fuse_ln", True)):
        y = fast_linear(owner, name, x, weight, bias)
        y = ops.row_ln(y.contiguous(), gamma, beta, eps, pre=None if pre is None else pre.contiguous().view_as(y),
                       post=None if post is None else post.contiguous().view_as(y), relu=relu)
        if out is not None:
            out.copy_(y)
            return out
        return y
    wp = _cached(owner, "wp_" + name, [weight], lambda: ops.pack_b_fragments(weight))
    return ops.slot_gemm_ln(x.contiguous(), wp, bias, gamma, beta, eps, pre=None if pre is None else pre.contiguous(),
                            post=None if post is None else post.contiguous(), relu=relu, out=out)


_CU_COUNT = {}


def _num_cus(device):
    idx = torch.device(device).index or 0
    if idx not in _CU_COUNT:
        _CU_COUNT[idx] = torch.cuda.get_device_properties(idx).multi_processor_count
    return _CU_COUNT[idx]


def fast_ffn(owner, x, lin1, lin2, norm, act, post=None):
    """LN(x + lin2(act(lin1(x)))) (+ post): the feed-forward block of a stage (dynamic_mask_head.py:379-385, :519-525). bf16 mode:
    ONE launch (csrc/slot_ffn.hip, the hidden tensor stays on the CU), bitwise the two K8 launches it replaces; otherwise those."""
    H, K = lin1.weight.shape
    split = "fp16" if getattr(owner, "precision", "bf16") == "fp16x2" else "bf16"
    if (getattr(owner, "precision", "bf16") == "fp32" or K != 256 or lin2.weight.shape[0] != 256 or H % 256 or not x.is_cuda
            or not owner.use_slot_gemm or not getattr(owner, "fuse_ffn", True) or not getattr(owner, "fuse_ln", True)
            or lin1.bias is None or lin2.bias is None):
        hid = fast_linear(owner, "linear1", x, lin1.weight, lin1.bias, act=act)
        return fast_linear_ln(owner, "linear2", hid, lin2.weight, lin2.bias, norm, pre=x, post=post)
    rows = x.numel() // K
    if -(-rows // 64) * 2 < _num_cus(x.device):
        # a workgroup of the one-launch form walks the whole hidden dimension for its 64 rows (~95 us): it pays once the launch
        # covers at least half the chip (measured: 16 000 rows 124 vs 179 us, 8 000 rows 97 vs 102, 500 rows 94 vs 47)
        hid = fast_linear(owner, "linear1", x, lin1.weight, lin1.bias, act=act)
        return fast_linear_ln(owner, "linear2", hid, lin2.weight, lin2.bias, norm, pre=x, post=post)
    sfx = "_f16" if split == "fp16" else ""
    w1 = _cached(owner, "wp_linear1" + sfx, [lin1.weight], lambda: ops.pack_b_fragments(lin1.weight, split))
    w2 = _cached(owner, "wp_linear2" + sfx, [lin2.weight], lambda: ops.pack_b_fragments(lin2.weight, split))
    xc = x.contiguous()
    return ops.slot_ffn(xc, w1, lin1.bias, w2, lin2.bias, norm.weight, norm.bias, norm.eps,
                        act={"relu": ops.ACT_RELU, "gelu": ops.ACT_GELU}[act], pre=xc,
                        post=None if post is None else post.contiguous())


FP16_MAX = 65504.0


def _act_name(fn):
    """Epilogue name of K8 for the FFN activation; anything else than relu / gelu (the reference also accepts "glu",
    dynamic_mask_head.py:575-583) has no fused form here and must not be dropped silently."""
    if fn is F.relu:
        return "relu"
    if fn is F.gelu:
        return "gelu"
    raise NotImplementedError(f"FFN activation {getattr(fn, '__name__', fn)!r}: the slot-side kernels implement relu and gelu (the released configs)")


class ConvModule(nn.Module):
    """1x1 conv + bias held as ``.conv`` - what the reference's ConvModule builds for
    activation=None and no norm (mmdet/models/utils/conv_module.py:95-97, :135)."""

    def __init__(self, in_channels, out_channels, kernel_size, padding=0, activation=None):
        super().__init__()
        if kernel_size != 1 or padding != 0 or activation is not None:
            raise NotImplementedError("only the 1x1 / no-activation form used by the slot head")
        self.conv = nn.Conv2d(in_channels, out_channels, 1, bias=True)

    def forward(self, x):
        return self.conv(x)


class MaskDynamicConv(nn.Module):
    """Retriever of the Panoptic Retriever (:403-461). forward keeps the reference signature
    (pro_features [N, L, C], features [N, C, H, W], pos [N, C, H, W]); ``forward_pm`` is the batched
    pixel-major entry the head uses."""

    def __init__(self, dh_dim=256, softmax_dim="slots"):
        super().__init__()
        if softmax_dim != "slots":
            raise NotImplementedError("the HIP kernel implements softmax over slots (the configured mode)")
        self.hidden_dim = dh_dim
        self.softmax_dim = softmax_dim
        self.to_q = nn.Linear(dh_dim, dh_dim, bias=True)
        self.to_k = nn.Linear(dh_dim, dh_dim, bias=True)
        self.to_v = nn.Linear(dh_dim, dh_dim, bias=True)
        self.norm_q = nn.LayerNorm(dh_dim)
        self.norm_k = nn.LayerNorm(dh_dim)
        self.norm_v = nn.LayerNorm(dh_dim)
        self.norm1 = nn.LayerNorm(dh_dim)
        self.activation = nn.ReLU(inplace=True)
        self.split_p = True     # carry softmax probabilities as bf16 hi+lo inside K1
        self.precision = DEFAULT_PRECISION   # "fp16x2" (default), "bf16" (16-bit level maps), "fp32" (exact mode, csrc/exact_f32.hip)
        # bf16 mode, two forms of the same function: "fused" (default where it applies, L <= 128): statistics-fused retriever
        # K3' + K1' - no k / v tensors, the map is read once per kernel; "kv": K3 writes bf16 k / v, K1 streams them
        self.retriever = "fused"
        self.use_slot_gemm = True   # K8 for the dense layers (bf16 mode)
        # The fused form converts the stored bf16 map to fp16 in LDS: exact for |f| in [6.1e-5, 65504]; smaller values lose bits
        # (down to 6e-8, then zero), larger ones SATURATE at +-65504 (v_cvt_pkrtz). range_check = True looks at the map's
        # maximum before every fused call (one device -> host sync) and runs the kv form - another HIP kernel pair, bf16
        # arithmetic, no fp16 staging - for a map that exceeds the range, with a one-time warning. Off by default: maps behind
        # the level-fusion conv of batch-normalised features are O(1) ... O(100).
        self.range_check = False
        # operands of the query-side products (to_q, the key fold, the position terms): "fp16" hi + lo (default), "bf16" hi + lo, "fp32" library
        self.query_side = "fp16"

    def _bf16_weights(self):
        """to_k / to_v weight matrices rounded to bf16 once (re-derived if the parameters change)."""
        key = (self.to_k.weight._version, self.to_v.weight._version, self.to_k.weight.data_ptr())
        if getattr(self, "_wcache_key", None) != key:
            self._wcache = (self.to_k.weight.detach().to(BF16).contiguous(),
                            self.to_v.weight.detach().to(BF16).contiguous())
            self._wcache_key = key
        return self._wcache

    def _fused_consts(self):
        """Weight-only constants of the fused form, derived once per weight version in float64 on the host:
        centred projections W~ = (I - 11^T/256) W, b~ = b - mean(b); the upper-triangular QR factor of [W~ | b~] for the
        statistics kernel; the slot-side matrices of the key fold and of the value epilogue."""
        srcs = [self.to_k.weight, self.to_k.bias, self.to_v.weight, self.to_v.bias, self.norm_k.weight, self.norm_k.bias,
                self.norm_v.weight, self.norm_v.bias]

        def build():
            dev = self.to_k.weight.device
            out = {}
            for name, lin in (("k", self.to_k), ("v", self.to_v)):
                w = lin.weight.detach().double().cpu()
                b = lin.bias.detach().double().cpu()
                wc = w - w.mean(dim=0, keepdim=True)                    # subtract the mean over OUTPUT channels (rows)
                bc = b - b.mean()
                r = torch.linalg.qr(torch.cat([wc, bc[:, None]], dim=1), mode="r").R          # [256, 257] upper trapezoidal
                # both factors as fp16 (csrc/retr_stats.hip, "Precision"); the float64 key factor stays on the host for the
                # position tables of retr_pos_tables()
                out["r" + name] = torch.triu(r[:, :256]).to(dev).to(torch.float16).contiguous()
                out["rb" + name] = r[:, 256].float().to(dev).contiguous()
                # reference precision (csrc/retr_stats_hl.hip): R = hi + lo, two fp16 matrices
                out["r" + name + "_lo"] = (torch.triu(r[:, :256]) - out["r" + name].double().cpu()).to(dev).to(torch.float16).contiguous()
                if name == "k":
                    out["rk64"] = torch.triu(r[:, :256])
                out["wc" + name], out["bc" + name] = wc, bc
            out["wck"] = out["wck"].float().to(dev).contiguous()                               # Q'' = (q * gamma_k) @ W~_k
            out["bck"] = out["bck"].float().to(dev).contiguous()
            gv = self.norm_v.weight.detach().double().cpu()
            wext = torch.zeros((272, 256), dtype=torch.float64)
            wext[:256] = (gv[:, None] * out.pop("wcv")).t()                                   # (gamma_v * W~_v)^T
            wext[256] = gv * out.pop("bcv")
            wext[257] = self.norm_v.bias.detach().double().cpu()
            out["wext"] = wext.float().to(dev).contiguous()
            out["wext_lin"] = out["wext"].t().contiguous()                                      # [256, 272]: the nn.Linear form of the same product
            out["wck_lin"] = out["wck"].t().contiguous()                                        # [256 (i), 256 (c)]
            return out
        return _cached(self, "fused", srcs, build)

    def retr_pos_tables(self, pos_tabs):
        """Position term of the key statistics as two fp32 tables (csrc/retr_stats.hip): Ty [H, 256] = ytab R_k[:, :128]^T,
        Tx [W, 256] = xtab R_k[:, 128:]^T, R_k the float64 QR factor. Derived once per (weights, level geometry)."""
        if pos_tabs is None:
            return None
        c = self._fused_consts()
        ytab, xtab = pos_tabs
        cache = c.setdefault("pos_tables", {})
        key = (ytab.data_ptr(), xtab.data_ptr(), tuple(ytab.shape), tuple(xtab.shape))
        if key not in cache:
            rk = c["rk64"]
            half = rk.shape[1] // 2
            ty = ytab.detach().double().cpu() @ rk[:, :half].t()
            tx = xtab.detach().double().cpu() @ rk[:, half:].t()
            cache[key] = (ty.float().to(ytab.device).contiguous(), tx.float().to(xtab.device).contiguous(), ytab, xtab)
        return cache[key][:2]

    def stats_hl_tables(self, pos_tabs):
        """Tables of csrc/retr_stats_hl.hip, columns in accumulator order: (Ty + r_k [H or 1, 256], Tx [W or 1, 256], r_v [256]) fp32 and
        whether Tx is in the tiled order (W % 32 == 0: ops.tile_tx_table). Derived once per (weights, level geometry)."""
        c = self._fused_consts()
        cache = c.setdefault("hl_tables", {})
        key = None if pos_tabs is None else (pos_tabs[0].data_ptr(), pos_tabs[1].data_ptr(), tuple(pos_tabs[0].shape), tuple(pos_tabs[1].shape))
        if key not in cache:
            perm = ops.acc_order_perm(c["rbk"].device)
            if pos_tabs is None:
                tyk = c["rbk"][perm][None].contiguous()
                txk = torch.zeros((1, perm.numel()), dtype=torch.float32, device=perm.device)
            else:
                ty, tx = self.retr_pos_tables(pos_tabs)
                tyk = (ty + c["rbk"][None])[:, perm].contiguous()
                txk = tx[:, perm].contiguous()
            tiled = pos_tabs is not None and txk.shape[0] % 32 == 0
            if tiled:
                txk = ops.tile_tx_table(txk)
            cache[key] = (tyk, txk, c["rbv"][perm].contiguous(), tiled, pos_tabs)
        return cache[key][:4]

    def stats_hl(self, feat_pm, hw, pos_tabs):
        """K3-HL (csrc/retr_stats_hl.hip): the aux rows of this stage on the fp16 hi + lo planes - factors AND map as hi + lo, three MFMAs per
        product. Depends on the map and the weights only (not on the slots): forward_clip may run it ahead on another stream."""
        c = self._fused_consts()
        tyk, txk, rbv_p, tiled = self.stats_hl_tables(pos_tabs)
        return ops.retr_stats_hl(feat_pm, hw[0], hw[1], tyk, txk, c["rk"], c["rk_lo"], self.norm_k.eps, c["rv"], c["rv_lo"], rbv_p,
                                 self.norm_v.eps, tx_tiled=tiled)

    def stats_args(self, pos_tabs):
        """(pos_proj, rk, rbk, eps_k, rv, rbv, eps_v): this stage's arguments of ops.retr_stats / one entry of ops.retr_stats_level."""
        c = self._fused_consts()
        return (self.retr_pos_tables(pos_tabs), c["rk"], c["rbk"], self.norm_k.eps, c["rv"], c["rbv"], self.norm_v.eps)

    def forward_fused(self, slots, feat_pm, hw, pos_tabs, stats=None):
        """K3' + K1' (csrc/retr_stats.hip, csrc/retr_attn.hip): slots [T, L, C] fp32, feat_pm [T, H*W, C] bf16.
        `stats` = the aux rows of ops.retr_stats if already computed for this (map, stage)."""
        c = self._fused_consts()
        T, L, C = slots.shape
        H, W = hw
        hl = feat_pm.dim() == 4                          # the map as fp16 hi + lo planes [2, T, HW, C] (precision "fp16x2")
        if self.norm_v.eps < 4e-6:
            # the kernels carry 2^7 * P * rstd_v as fp16 (csrc/common.h, kPScale): rstd_v <= 1 / sqrt(eps_v) must stay below 511
            raise ValueError(f"fused retriever: norm_v.eps = {self.norm_v.eps} < 4e-6 is outside the fp16 range of the probabilities; "
                             "use set_mode('bf16_kv') or set_mode('fp32')")
        if stats is None:
            # statistics already computed for this map by the level pass (MultiScaleDynamicMaskHead.forward_clip)?
            pending = getattr(self, "_level_stats", None)
            self._level_stats = None
            if pending is not None and pending[0] is feat_pm:
                stats = pending[1]
                if len(pending) > 2:                     # computed on the pixel-side stream (forward_clip): this stream continues behind it
                    torch.cuda.current_stream(feat_pm.device).wait_event(pending[2])
        if hl:
            if stats is None:
                stats = self.stats_hl(feat_pm, hw, pos_tabs)
        elif stats is None:
            stats = ops.retr_stats(feat_pm, H, W, *self.stats_args(pos_tabs))
        LP = ops.retr_slot_pad(L)
        # :431 q = norm_q(to_q(slots)); g = q * gamma_k (zero rows up to LP), c3 = q . beta_k, a1 = g . b~_k: one launch
        # The query side ends up inside logits that are sums of 256 terms of magnitude ~5 cancelling to <= 80: bf16 hi + lo operands
        # (16 bits, ~1e-5 relative on Q'') cost ~5e-4 on the slot update. query_side = "fp16" (default): the same K8 / K9 launches
        # with fp16 hi + lo operands (22 bits; LayerNorm outputs, O(0.1) weights and sine tables are well inside fp16's range);
        # "bf16": round 2's operands; "fp32": the GEMM library.
        qs = self.query_side if self.use_slot_gemm else "fp32"
        if qs == "fp32":
            xq = F.linear(slots, self.to_q.weight, self.to_q.bias)
        else:
            xq = fast_linear(self, "to_q", slots, self.to_q.weight, self.to_q.bias, split=qs)
        gp, c3, a1 = ops.retr_query_prep(xq.contiguous(), self.norm_q.weight, self.norm_q.bias, self.norm_q.eps, self.norm_k.weight,
                                         self.norm_k.bias, c["bck"], LP)
        # Q'' [T, LP, 256] = gp @ W~_k: the key projection folded into the queries
        q2 = F.linear(gp, c["wck_lin"]).contiguous() if qs == "fp32" else fast_linear(self, "wck", gp, c["wck_lin"], split=qs)
        qh, ql = ops.retr_split(q2)
        if pos_tabs is not None:                                           # separable position terms + a' (two small tables per frame)
            ytab, xtab = pos_tabs
            # K9: cy[t, y, l] = a'[t, l] + ytab[y] . Q''[t, l, :128], cx[t, x, l] = xtab[x] . Q''[t, l, 128:] (shared tables: batch stride 0)
            if qs == "fp32":
                cy = (torch.matmul(ytab, q2[:, :, :C // 2].transpose(1, 2)) + a1[:, None, :]).contiguous()
                cx = torch.matmul(xtab, q2[:, :, C // 2:].transpose(1, 2)).contiguous()
            else:
                cy = ops.bgemm(ytab, q2[:, :, :C // 2], bias=a1, split=qs)
                cx = ops.bgemm(xtab, q2[:, :, C // 2:], split=qs)
        else:
            cy = a1[:, None, :].expand(T, H, LP).contiguous()
            cx = torch.zeros((T, W, LP), dtype=torch.float32, device=slots.device)
        if hl:
            ext = ops.retr_attn_hl(qh, ql, cy, cx, c3, feat_pm, stats, L, H, W)
            # the pixel sums A_l reach |f| * (pixels a slot owns): beyond fp16's range at the fine levels, so this one product (272 -> 256 on
            # [T L] rows) runs in the library's fp32; norm1 + ReLU on K5
            return ops.row_ln(F.linear(ext, c["wext_lin"]), self.norm1.weight, self.norm1.bias, self.norm1.eps, relu=True)
        ext = ops.retr_attn(qh, ql, cy, cx, c3, feat_pm, stats, L, H, W)
        # :456 (value projection after the sum) + :458-459 (norm1, ReLU) in one launch
        return fast_linear_ln(self, "wext", ext, c["wext_lin"], None, self.norm1, relu=True)

    def project_kv(self, feat_pm, hw, pos_tabs):
        """K3: feat_pm [T, H*W, C] bf16, hw = (H, W), pos_tabs = (ytab, xtab) or None -> k, v bf16."""
        wk, wv = self._bf16_weights()
        return ops.kv_project(feat_pm, hw[0], hw[1], pos_tabs, wk, self.to_k.bias, self.norm_k.weight,
                              self.norm_k.bias, self.norm_k.eps, wv, self.to_v.bias, self.norm_v.weight,
                              self.norm_v.bias, self.norm_v.eps)

    def forward_pm(self, slots, feat_pm, hw, pos_tabs):
        """slots [T, L, C] fp32, feat_pm [T, H*W, C] bf16 (fp32 in exact mode) -> [T, L, C] fp32 (K3 then K1)."""
        if self.precision == "fp32":
            q = ops.row_ln(self.to_q(slots), self.norm_q.weight, self.norm_q.bias, self.norm_q.eps)
            wkT = _cached(self, "wkT", [self.to_k.weight], lambda: self.to_k.weight.t().contiguous())
            wvT = _cached(self, "wvT", [self.to_v.weight], lambda: self.to_v.weight.t().contiguous())
            k, v = ops.kv_project_f32(feat_pm, hw[0], hw[1], pos_tabs, wkT, self.to_k.bias, self.norm_k.weight,
                                      self.norm_k.bias, self.norm_k.eps, wvT, self.to_v.bias, self.norm_v.weight,
                                      self.norm_v.bias, self.norm_v.eps)
            return ops.slot_attn_f32(q, k, v, self.norm1.weight, self.norm1.bias, eps=self.norm1.eps)
        if self.precision == "fp16x2":
            if feat_pm.dim() != 4:
                raise ValueError("precision 'fp16x2' takes the level map as fp16 hi + lo planes [2, T, HW, C] (ops.split_hl / ops.level_fuse_hl)")
            return self.forward_fused(slots, feat_pm, hw, pos_tabs)
        if feat_pm.dtype == torch.float16 and self.retriever != "fused":
            raise NotImplementedError("fp16 level maps (map_dtype='fp16') go with the fused retriever")
        if self.retriever == "fused":
            if self.range_check and feat_pm.dtype != torch.float16 and float(feat_pm.abs().max()) > FP16_MAX:
                if not getattr(MaskDynamicConv, "_warned_range", False):
                    MaskDynamicConv._warned_range = True
                    import warnings
                    warnings.warn("fused retriever: the feature map exceeds the fp16 range (|f| > 65504); running the kv form "
                                  "(bf16 k / v tensors, csrc/kv_project.hip + csrc/slot_attn.hip) for such maps")
                self._level_stats = None
            else:
                return self.forward_fused(slots, feat_pm, hw, pos_tabs)
        q = ops.row_ln(self.to_q(slots), self.norm_q.weight, self.norm_q.bias, self.norm_q.eps, out_bf16=True)
        k, v = self.project_kv(feat_pm, hw, pos_tabs)
        return ops.slot_attn(q, k, v, self.norm1.weight, self.norm1.bias, eps=self.norm1.eps, split_p=self.split_p)

    def forward(self, pro_features, features, pos, gt_non_void_mask=None):
        assert gt_non_void_mask is None
        n, c, h, w = features.shape
        if self.precision == "fp16x2":
            feat_pm = ops.split_hl(features.permute(0, 2, 3, 1).reshape(n, h * w, c).float().contiguous())
            return self.forward_pm(pro_features.float(), feat_pm, (h, w), pos_tables_from_map(pos))
        store = torch.float32 if self.precision == "fp32" else (torch.float16 if getattr(self, "map_dtype", "bf16") == "fp16" else BF16)
        feat_pm = features.permute(0, 2, 3, 1).reshape(n, h * w, c).to(store).contiguous()
        return self.forward_pm(pro_features.float(), feat_pm, (h, w), pos_tables_from_map(pos))


def pos_tables_from_map(pos):
    """[N, C, H, W] position map -> the separable (ytab [H, C/2], xtab [W, C/2]) K3 consumes. The sine
    embedding is separable by construction (position_encoding.py:251-255); anything else is refused."""
    if pos is None:
        return None
    p = pos[0].float()
    half = p.shape[0] // 2
    ytab = p[:half, :, 0].t().contiguous()
    xtab = p[half:, 0, :].t().contiguous()
    if not (torch.equal(p[:half], ytab.t()[:, :, None].expand_as(p[:half]))
            and torch.equal(p[half:], xtab.t()[:, None, :].expand_as(p[half:]))):
        raise NotImplementedError("only separable (sine) position embeddings are supported by the HIP path")
    return ytab, xtab


class SlotsDynamicConv(nn.Module):
    """Retriever of the Video Retriever (:530-572): the same math among the T*L slots themselves
    (<= 2000 rows) - far below one workgroup's worth of work, left to PyTorch."""

    def __init__(self, dh_dim=256, softmax_dim="slots"):
        super().__init__()
        if softmax_dim != "slots":
            raise NotImplementedError
        self.hidden_dim = dh_dim
        self.softmax_dim = softmax_dim
        self.to_q = nn.Linear(dh_dim, dh_dim, bias=True)
        self.to_k = nn.Linear(dh_dim, dh_dim, bias=True)
        self.to_v = nn.Linear(dh_dim, dh_dim, bias=True)
        self.norm_q = nn.LayerNorm(dh_dim)
        self.norm_k = nn.LayerNorm(dh_dim)
        self.norm_v = nn.LayerNorm(dh_dim)
        self.norm1 = nn.LayerNorm(dh_dim)
        self.activation = nn.ReLU(inplace=True)
        self.precision = DEFAULT_PRECISION
        self.use_slot_gemm = True

    def forward(self, curr_features, features, pos, groups=1):
        """`groups` > 1: the rows are `groups` independent clips of equal length laid end to end; attention stays
        inside each clip (a batch of clips per launch - the reference handles one clip per call)."""
        if pos is not None or curr_features is not features:
            assert groups == 1
            q = self.norm_q(self.to_q(curr_features))
            k = self.norm_k(self.to_k(features if pos is None else features + pos))
            v = self.norm_v(self.to_v(features))
        else:
            # the call site of the head (q, k, v all from the same T*L slots): one batched GEMM for the
            # three projections and one K5 launch for the three LayerNorms
            x = curr_features.reshape(-1, self.hidden_dim)
            lins, norms = (self.to_q, self.to_k, self.to_v), (self.norm_q, self.norm_k, self.norm_v)
            w3 = _cached(self, "w3", [m.weight for m in lins], lambda: torch.stack([m.weight.t() for m in lins]).contiguous())
            b3 = _cached(self, "b3", [m.bias for m in lins], lambda: torch.stack([m.bias for m in lins]).unsqueeze(1).contiguous())
            g3 = _cached(self, "g3", [m.weight for m in norms], lambda: torch.stack([m.weight for m in norms]).contiguous())
            e3 = _cached(self, "e3", [m.bias for m in norms], lambda: torch.stack([m.bias for m in norms]).contiguous())
            if self.precision != "fp32" and self.use_slot_gemm and x.is_cuda:
                qkv = torch.empty((3,) + tuple(x.shape), dtype=torch.float32, device=x.device)
                for i, (m, nm) in enumerate(zip(lins, norms)):                      # three K8 launches (projection + its LayerNorm) into one [3, M, C] buffer
                    fast_linear_ln(self, f"qkv{i}", x, m.weight, m.bias, nm, out=qkv[i])
            else:
                qkv = torch.baddbmm(b3, x.unsqueeze(0).expand(3, -1, -1), w3)        # [3, M, C]
                qkv = ops.row_ln(qkv, g3, e3, self.norm_q.eps, rows_per_group=x.shape[0])
            q, k, v = (qkv[i].view(groups, -1, self.hidden_dim) for i in range(3))
        # softmax over the QUERY axis (dim=1 of [1, Lq, Lk], :562) = last-dim softmax of the transposed logits
        if self.precision != "fp32" and self.use_slot_gemm and q.is_cuda:
            # K9 (csrc/bgemm.hip): both products on the matrix cores in split bf16; the second one reads attn_t k-major
            sp = "fp16" if self.precision == "fp16x2" else "bf16"             # operand split of K9 (fp16 hi + lo: 22 bits)
            # fp16 split: the probabilities travel times 2^14 (<= 16 384). The softmax runs over the QUERY axis, so a query can receive
            # almost no mass from any key - its whole output row would sit in fp16's subnormal range (absolute resolution 6e-8) while
            # norm1 behind it scales the row back up by up to 1 / sqrt(eps) = 316 (measured: 2e-4 on a stage, against 1e-5)
            psc = 16384.0 if sp == "fp16" else 1.0
            attn_t = ops.row_softmax(ops.bgemm(k, q, split=sp), inplace=True, scale=psc)            # [G, Lk, Lq]
            out = ops.bgemm(attn_t.transpose(1, 2), v.transpose(1, 2), split=sp, alpha=1.0 / psc).view(1, -1, self.hidden_dim)
        else:
            attn_t = torch.softmax(k @ q.transpose(-1, -2), dim=-1)     # [1, Lk, Lq]
            out = (attn_t.transpose(-1, -2) @ v).reshape(1, -1, self.hidden_dim)
        return ops.row_ln(out.contiguous(), self.norm1.weight, self.norm1.bias, self.norm1.eps, relu=True)


@HEADS.register_module
class TemporalSlotsHead(nn.Module):
    """Video Retriever (:464-527)."""

    def __init__(self, d_model, dim_feedforward=2048, dropout=0.1, activation="relu", softmax_dim="slots",
                 drop_path=0.):
        super().__init__()
        if dropout != 0.0 or drop_path != 0.0:
            raise NotImplementedError("inference path: dropout / drop_path must be 0")
        self.d_model = d_model
        self.inst_interact = SlotsDynamicConv(dh_dim=d_model, softmax_dim=softmax_dim)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)   # present in checkpoints, unused by the reference too
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.activation = _get_activation_fn(activation)
        self.precision = DEFAULT_PRECISION
        self.use_slot_gemm = True

    def forward(self, features, mask_query, pos=None, query_pos=None, add_input=False, groups=1):
        """:494-527. add_input=True additionally returns mask_query + result (the caller's residual, :317)
        fused into the last K5 launch. groups: independent clips in the row dimension (see SlotsDynamicConv)."""
        assert query_pos is None
        x = mask_query.view(1, -1, self.d_model)
        f = x if features is mask_query else features.view(1, -1, self.d_model)
        r = self.inst_interact(x, f, pos, groups=groups)
        u = ops.row_ln(r, self.norm2.weight, self.norm2.bias, self.norm2.eps, pre=x.contiguous())          # :515-517
        out = fast_ffn(self, u, self.linear1, self.linear2, self.norm3, _act_name(self.activation),
                       post=x.contiguous() if add_input else None)                                            # :519-520, :524-525
        return out.squeeze(0)


class MaskRCNNHead(nn.Module):
    """One stage (:231-400)."""

    def __init__(self, d_model, num_classes, dim_feedforward=2048, nhead=8, dropout=0.1, activation="relu",
                 scale_clamp=math.log(100000.0 / 16), num_cls=1, num_reg=3, use_focal=True, softmax_dim="slots",
                 drop_path=0., temporal_query_attention_config=None):
        super().__init__()
        if dropout != 0.0 or drop_path != 0.0:
            raise NotImplementedError("inference path: dropout / drop_path must be 0")
        self.d_model = d_model
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.inst_interact = MaskDynamicConv(dh_dim=d_model, softmax_dim=softmax_dim)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.activation = _get_activation_fn(activation)
        self.drop_path = None
        self.temporal_query_head = None
        if temporal_query_attention_config is not None:
            self.temporal_query_head = TemporalSlotsHead(**temporal_query_attention_config)
        cls_module, reg_module = [], []
        for _ in range(num_cls):
            cls_module += [nn.Linear(d_model, d_model, False), nn.LayerNorm(d_model), nn.ReLU(inplace=True)]
        for _ in range(num_reg):
            reg_module += [nn.Linear(d_model, d_model, False), nn.LayerNorm(d_model), nn.ReLU(inplace=True)]
        self.cls_module = nn.ModuleList(cls_module)
        self.reg_module = nn.ModuleList(reg_module)
        self.use_focal = use_focal
        self.class_logits = nn.Linear(d_model, num_classes)
        self.scale_clamp = scale_clamp
        self.precision = DEFAULT_PRECISION
        self.use_slot_gemm = True

    def _self_attention(self, slots, residual_norm=False):
        """residual_norm: return norm1(slots + attention) (:356-358) instead of the attention output - in bf16 mode the
        LayerNorm then rides in the epilogue of the output projection.
        nn.MultiheadAttention(self_attn)(x, x, x) for frames-as-batch slots [T, L, C] (:346-355), with the module's own
        parameters but without its sequence-first layout: the packed projection runs on the contiguous [T*L, C] rows and
        the attention kernel reads q / k / v as strided views of its output ([T, L, heads, 32] is the layout that kernel
        uses internally), so no transposed copy is made on the way in or out."""
        mha = self.self_attn
        T, L, C = slots.shape
        nh = mha.num_heads
        qkv = fast_linear(self, "in_proj", slots, mha.in_proj_weight, mha.in_proj_bias).view(T, L, 3, nh, C // nh)
        q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))          # [T, heads, L, C / heads] views
        if self.precision == "fp16x2" and C // nh == 32 and L <= 256 and self.use_slot_gemm:
            # the library's attention kernel with fp16 hi + lo operands (22 bits; round 5: three GEMM-library launches + softmax before)
            o = ops.slot_self_attn(qkv.view(T, L, 3 * C), nh, split="fp16")
            if residual_norm:
                return fast_linear_ln(self, "out_proj", o, mha.out_proj.weight, mha.out_proj.bias, self.norm1, pre=slots)
            return fast_linear(self, "out_proj", o, mha.out_proj.weight, mha.out_proj.bias)
        if self.precision in ("fp32", "fp16x2"):                            # explicit fp32 products, torch's own order (:352)
            attn = torch.softmax((q * (1.0 / math.sqrt(C // nh))) @ k.transpose(-1, -2), dim=-1)
            o = attn @ v
            if self.precision == "fp16x2":                                  # the output projection (+ residual + norm1) on K8 / K5
                o = o.transpose(1, 2).reshape(T, L, C)
                if residual_norm:
                    return fast_linear_ln(self, "out_proj", o, mha.out_proj.weight, mha.out_proj.bias, self.norm1, pre=slots)
                return fast_linear(self, "out_proj", o, mha.out_proj.weight, mha.out_proj.bias)
        elif C // nh == 32 and L <= 256:
            # the library's own kernel for these tiny (L x L x 32 per head) problems, on the packed projection as it stands
            o = ops.slot_self_attn(qkv.view(T, L, 3 * C), nh)
            if residual_norm:
                return fast_linear_ln(self, "out_proj", o, mha.out_proj.weight, mha.out_proj.bias, self.norm1, pre=slots)
            return fast_linear(self, "out_proj", o, mha.out_proj.weight, mha.out_proj.bias)
        else:
            o = F.scaled_dot_product_attention(q, k, v)                     # softmax(q k^T / sqrt(C / heads)) v
        a = F.linear(o.transpose(1, 2).reshape(T, L, C), mha.out_proj.weight, mha.out_proj.bias)
        if residual_norm:
            return ops.row_ln(a, self.norm1.weight, self.norm1.bias, self.norm1.eps, pre=slots)
        return a

    def forward_till_ffn_pm(self, slots, feat_pm, hw, pos_tabs):
        """:342-388 for all frames at once. slots [T, L, C] fp32 contiguous."""
        x1 = self._self_attention(slots, residual_norm=True)                                              # :346-358
        r = self.inst_interact.forward_pm(x1, feat_pm, hw, pos_tabs)                                        # :368
        x2 = ops.row_ln(r, self.norm2.weight, self.norm2.bias, self.norm2.eps, pre=x1)                     # :374-376
        return fast_ffn(self, x2, self.linear1, self.linear2, self.norm3, _act_name(self.activation))       # :379, :384-385

    def forward_after_ffn_pm(self, obj, out_cls=None, out_emb=None):
        """:390-400 -> (class_logits [T, L, nc], slot embedding [T, L, C]). The class and the embedding tower
        have the same shape, so layer i of both runs as one batched GEMM + one K5 launch. out_cls / out_emb: contiguous fp32
        buffers the two results are written into by the kernels that produce them (the clip entry stacks the stages' outputs)."""
        T, L, C = obj.shape
        if len(self.cls_module) != len(self.reg_module):
            c = r = obj
            for layer in self.cls_module:
                c = layer(c)
            for layer in self.reg_module:
                r = layer(r)
            return self.class_logits(c), r
        n_cls, n_reg = len(self.cls_module) // 3, len(self.reg_module) // 3
        lins = [self.cls_module[3 * k] for k in range(n_cls)] + [self.reg_module[3 * k] for k in range(n_reg)]
        if (self.precision == "bf16" and self.use_slot_gemm and obj.is_cuda and getattr(self, "fuse_ln", True)
                and n_cls >= 1 and n_reg >= 1 and n_cls + n_reg <= 6 and -(-T * L // 64) * 2 >= _num_cus(obj.device)
                and all(isinstance(m, nn.Linear) and m.bias is None and tuple(m.weight.shape) == (256, 256) for m in lins)):
            # both towers in ONE launch (csrc/slot_chain.hip): each tower's layers chained on the CU, both off the same input tile;
            # bitwise the per-layer K8 launches; 47 against 57 us at 16 000 rows (slower below half a chip: the per-layer form stays)
            def layer(name, mods, k, **kw):
                lin, norm = mods[3 * k], mods[3 * k + 1]
                wp = _cached(self, f"wp_{name}{3 * k}", [lin.weight], lambda: ops.pack_b_fragments(lin.weight))
                return dict(wpack=wp, gamma=norm.weight, beta=norm.bias, eps=norm.eps, relu=True, src="x" if k == 0 else "prev", **kw)
            emb = out_emb.view(T * L, C) if out_emb is not None else torch.empty((T * L, C), dtype=torch.float32, device=obj.device)
            ctmp = torch.empty((T * L, C), dtype=torch.float32, device=obj.device)
            chain = [layer("reg", self.reg_module, k, **(dict(out=emb) if k == n_reg - 1 else {})) for k in range(n_reg)]
            chain += [layer("cls", self.cls_module, k, **(dict(out=ctmp) if k == n_cls - 1 else {})) for k in range(n_cls)]
            ops.slot_chain(obj.reshape(T * L, C), chain)
            nc_ = self.class_logits.weight.shape[0]
            cls = ops.bgemm(ctmp, self.class_logits.weight, bias=self.class_logits.bias,
                            out=None if out_cls is None else out_cls.view(1, T * L, nc_)).view(T, L, -1)
            return cls, emb.view(T, L, C)
        x = obj.reshape(1, T * L, C).expand(2, -1, -1)
        last = len(self.cls_module) - 3
        for i in range(0, len(self.cls_module), 3):
            lc, lr = self.cls_module[i], self.reg_module[i]
            nc, nr = self.cls_module[i + 1], self.reg_module[i + 1]
            w2 = _cached(self, f"tw{i}", [lc.weight, lr.weight], lambda: torch.stack([lc.weight.t(), lr.weight.t()]).contiguous())
            g2 = _cached(self, f"tg{i}", [nc.weight, nr.weight], lambda: torch.stack([nc.weight, nr.weight]).contiguous())
            e2 = _cached(self, f"te{i}", [nc.bias, nr.bias], lambda: torch.stack([nc.bias, nr.bias]).contiguous())
            if self.precision != "fp32" and self.use_slot_gemm:
                if i == last and out_emb is not None:
                    y2 = (torch.empty((T * L, C), dtype=torch.float32, device=obj.device), out_emb.view(T * L, C))
                else:
                    y2 = torch.empty((2, T * L, C), dtype=torch.float32, device=obj.device)
                fast_linear_ln(self, f"cls{i}", x[0], lc.weight, None, nc, relu=True, out=y2[0])           # :394-397, layer + LayerNorm + ReLU per launch
                fast_linear_ln(self, f"reg{i}", x[1], lr.weight, None, nr, relu=True, out=y2[1])
                x = y2
            else:
                x = ops.row_ln(torch.bmm(x, w2), g2, e2, nc.eps, relu=True, rows_per_group=T * L)          # :394-397
        if self.precision != "fp32" and self.use_slot_gemm and x[0].is_cuda:         # :398 on K9 (20 columns: not a K8 shape)
            nc_ = self.class_logits.weight.shape[0]
            cls = ops.bgemm(x[0], self.class_logits.weight, bias=self.class_logits.bias,
                            out=None if out_cls is None else out_cls.view(1, T * L, nc_),
                            split="fp16" if self.precision == "fp16x2" else "bf16").view(T, L, -1)
        else:
            cls = self.class_logits(x[0].reshape(T, L, C))
        return cls, x[1].reshape(T, L, C)

    def forward_pm(self, slots, feat_pm, hw, pos_tabs, stage_enable, clips=1, out_cls=None, out_emb=None):
        T, L, C = slots.shape
        obj = self.forward_till_ffn_pm(slots.contiguous(), feat_pm, hw, pos_tabs)
        if stage_enable:
            flat = obj.reshape(T * L, C)                                    # concat along the slot axis (:310)
            obj = self.temporal_query_head(features=flat, mask_query=flat, add_input=True,
                                           groups=clips).reshape(T, L, C)                                     # :313-322
        else:
            assert self.temporal_query_head is None
        return self.forward_after_ffn_pm(obj, out_cls, out_emb)

    def forward(self, features, mask_query, pad_mask, pos=None, query_pos=None, gt_non_void_mask=None,
                stage_enable=True):
        """Reference signature: lists over frames of [1, C, H, W] / [1, L, C]."""
        assert pad_mask is None and query_pos is None and gt_non_void_mask is None
        T = len(features)
        _, c, h, w = features[0].shape
        store = torch.float32 if self.precision in ("fp32", "fp16x2") else (torch.float16 if getattr(self, "map_dtype", "bf16") == "fp16" else BF16)
        feat_pm = torch.cat(features, 0).permute(0, 2, 3, 1).reshape(T, h * w, c).to(store).contiguous()
        if self.precision == "fp16x2":
            feat_pm = ops.split_hl(feat_pm)
        tabs = pos_tables_from_map(pos[0]) if pos is not None else None
        logits, emb = self.forward_pm(torch.cat(mask_query, 0).float(), feat_pm, (h, w), tabs, stage_enable)
        return [logits[t:t + 1] for t in range(T)], [emb[t:t + 1] for t in range(T)], None, None


@HEADS.register_module
class MultiScaleDynamicMaskHead(nn.Module):
    """:36-229. Constructor arguments, attribute names and the forward contract follow the reference."""

    def __init__(self, dh_dim=256, num_classes=9, dim_feedforward=2048, nhead=8, dropout=0.0, activation="relu",
                 dh_num_heads=8, per_dh_num_heads=2, feat_num_levels=4, merge_operation="add", trans_in_dim=128,
                 return_intermediate=True, use_focal=True, prior_prob=0.01, num_cls=1, num_reg=3,
                 softmax_dim="slots", drop_path=0., temporal_query_attention_config=None,
                 apply_temporal_query_atten_stages=None, other_config=None):
        super().__init__()
        if not isinstance(per_dh_num_heads, (list, tuple)):
            assert per_dh_num_heads * feat_num_levels == dh_num_heads
            per_dh_num_heads = [per_dh_num_heads] * feat_num_levels
        else:
            assert sum(per_dh_num_heads) == dh_num_heads
        if merge_operation != "concat":
            raise NotImplementedError("the released configs use merge_operation='concat'")
        self.per_dh_num_heads = list(per_dh_num_heads)
        self.dh_dim = dh_dim
        # a config selects the mode (MODES below) with other_config=dict(mode="fp16x2"); applied once the stages exist
        self.map_dtype = "bf16"
        self.map_encoding = "auto"                       # "bf16": plain bf16 tensors for the bf16 policy (see _map_form)
        self.stats_form = "level"                        # "level": K3'' - both stages of a pyramid level from one read of the map; "stage": K3' per stage
        self.fuse_orders_in_one_launch = True            # fp16x2: all orders G^(m) of a level from one launch of K4-HL (False: one launch per order)
        self._cfg_mode = other_config.get("mode") if isinstance(other_config, dict) else None
        if isinstance(other_config, dict):
            # the switches of rounds 1 - 4 were collapsed into `mode`: a config that still carries them must not silently run another mode
            legacy = sorted(k for k in ("precision", "map_dtype", "retriever", "statistics") if k in other_config)
            if legacy:
                raise ValueError(f"other_config keys {legacy} are no longer read: select the head mode with other_config=dict(mode=...), "
                                 f"one of {sorted(self.MODES)} (default {DEFAULT_MODE!r})")
        self.trans_in_dim = trans_in_dim
        self.apply_temporal_query_atten_stages = apply_temporal_query_atten_stages
        self.other_config = other_config

        def make_stage(temporal_cfg):
            return MaskRCNNHead(d_model=dh_dim, num_classes=num_classes, dim_feedforward=dim_feedforward,
                                nhead=nhead, dropout=dropout, activation=activation, num_cls=num_cls,
                                num_reg=num_reg, softmax_dim=softmax_dim, drop_path=drop_path,
                                temporal_query_attention_config=temporal_cfg)
        stage_idx = 0
        for i in range(feat_num_levels):
            # a level gets the temporal sub-head iff its FIRST stage index is a temporal stage (:83-106)
            temporal = (apply_temporal_query_atten_stages is None
                        or stage_idx in apply_temporal_query_atten_stages)
            proto = make_stage(temporal_query_attention_config if temporal else None)
            setattr(self, f"head_series_{i}", _get_clones(proto, self.per_dh_num_heads[i]))
            stage_idx += self.per_dh_num_heads[i]
        self.conv_trans = ConvModule(trans_in_dim, dh_dim, 1, padding=0, activation=None)
        self.return_intermediate = return_intermediate
        self.feat_num_levels = feat_num_levels
        self.merge_operation = merge_operation
        self.use_focal = use_focal
        self.num_classes = num_classes
        if use_focal:
            self.prior_prob = prior_prob
            self.bias_value = -math.log((1 - prior_prob) / prior_prob)
        self.precision = DEFAULT_PRECISION
        self.mode = DEFAULT_MODE
        self._reset_parameters()
        self.set_mode(self._cfg_mode if self._cfg_mode is not None else DEFAULT_MODE)

    # THE mode surface of the head: name -> (precision, storage of the level maps, retriever form). What each costs and how far it sits
    # from the reference's own fp32 outputs at BASELINE's sizes: tests/test_full_size_gpu.py, bench.py (config.mode_*), README.
    #   "fp16x2"  the reference's precision ON THE MATRIX CORES: every 16-bit matrix operand as fp16 hi + lo (22 bits, three MFMAs per
    #             product), level maps as two fp16 planes (csrc/level_fuse_hl.hip), statistics / retriever / decode in their HL forms, slot
    #             side on K8 / K9 with fp16 hi + lo operands. Meets the north star's tolerance (1e-4 on the mask logits, identical slot argmax
    #             wherever decidable) free-running at full size - the mode bench.py's `value` is quoted on. |f| < 65 504, L <= 256.
    #   "fp32"    exact mode: fp32 storage and fp32 vector-ALU arithmetic everywhere (csrc/exact_f32.hip), the reference's own dtype
    #             (vps_temporal_slots.py:55); meets the tolerance as well, ~14x slower than fp16x2.
    #   "bf16"    BASELINE.json's storage policy: bf16 level maps (stored in the fp16 encoding where every consumer takes it), statistics-
    #             fused retriever K3'' + K1' with single fp16 operands. 2.5x the speed of fp16x2; the storage rounding of the maps alone moves
    #             the mask logits by ~1e-3, i.e. it does NOT meet the tolerance.
    #   "fp16"    the same kernels with fp16 level maps (same bytes, three more mantissa bits): ~1.5e-4 from the map rounding alone.
    #   "bf16_kv" round 1's form through bf16 k / v tensors (csrc/kv_project.hip + csrc/slot_attn.hip): the fallback for maps outside
    #             fp16's range (MaskDynamicConv.range_check) and the only form without fp16 staging.
    MODES = {
        "fp16x2": ("fp16x2", "bf16", "fused"),
        "fp32": ("fp32", "bf16", "fused"),
        "bf16": ("bf16", "bf16", "fused"),
        "fp16": ("bf16", "fp16", "fused"),
        "bf16_kv": ("bf16", "bf16", "kv"),
    }

    def set_mode(self, name):
        if name not in self.MODES:
            raise ValueError(f"mode must be one of {sorted(self.MODES)}, not {name!r}")
        prec, maps, retr = self.MODES[name]
        for m in self.modules():
            if hasattr(m, "precision"):
                m.precision = prec
                m.map_dtype = maps                       # the stages' reference-signature entry points store their maps the same way
            if hasattr(m, "retriever"):
                m.retriever = retr
        self.map_dtype = maps
        self.mode = name
        return self

    def set_slot_gemm(self, on):
        """Debug switch: K8 (split matrix-core GEMM, default) or the GEMM library in fp32 for the dense slot-side layers."""
        for m in self.modules():
            if hasattr(m, "use_slot_gemm"):
                m.use_slot_gemm = bool(on)
        return self

    def _reset_parameters(self):
        for p in self.parameters():                      # :127-136
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
            if self.use_focal and p.shape[-1] == self.num_classes:
                nn.init.constant_(p, self.bias_value)

    # ------------------------------------------------------------------------------------------
    def _map_form(self, cur=None):
        """How the fused level maps are stored: "fp16" (map_dtype "fp16"), "bf16" (bf16 tensors) or "bf16_in_fp16" - the bf16 storage
        policy (every rounding point rounds to bf16: the same values) in the fp16 ENCODING, which the matrix instructions of every
        consumer take directly, so that the bf16 -> fp16 pass over each tile in LDS disappears from K1' / K3'' (11 % of K1'). Chosen
        whenever every consumer is such a kernel: fused retriever, eight-wave form, fp32 NCHW incoming maps, no range_check (which has
        to see the unsaturated map); map_encoding = "bf16" forces plain bf16 tensors."""
        if getattr(self, "map_dtype", "bf16") == "fp16":
            return "fp16"
        mdcs = [m for m in self.modules() if hasattr(m, "retriever")]
        if self.map_encoding == "bf16" or any(m.retriever != "fused" or m.range_check or m.norm_v.eps < 4e-6 for m in mdcs):
            return "bf16"
        return "bf16_in_fp16"

    def _conv_weights(self, form=None, pre=None, level0=False):
        """K4's weight [256, 384] and bias in the operand type of `form`. pre = (W_t [128, 128], b_t [128] or None): a linear 1x1 map
        x = W_t y + b_t in front of the head (the detector's conv_trans, vps_capsule.py:76-79) FOLDED into them, so that K4 reads y
        itself: the x columns become W_x W_t (composed in float64), the bias b + W_x b_t. The shared conv then has two forms: a level
        with a previous map multiplies x by columns 256 .. 383 only, level 0 (cat(x, x, x), :171-176) by all three blocks."""
        conv = self.conv_trans.conv
        form = form or self._map_form()
        key = (conv.weight._version, conv.weight.data_ptr(), conv.bias._version, conv.bias.data_ptr(), form, bool(level0)) + \
            (() if pre is None else tuple((t.data_ptr(), t._version) for t in pre if t is not None))
        cache = self.__dict__.setdefault("_cw_cache", {})
        ent = cache.get(key)
        if ent is None:
            w = conv.weight.detach().reshape(self.dh_dim, self.trans_in_dim)
            b = conv.bias
            if pre is not None:
                wt = pre[0].detach().reshape(pre[0].shape[0], -1).double()
                if wt.shape[0] != self.trans_in_dim - self.dh_dim:
                    raise ValueError(f"pre_linear maps to {wt.shape[0]} channels, K4's incoming map has {self.trans_in_dim - self.dh_dim}")
                if wt.shape[1] != wt.shape[0]:
                    raise NotImplementedError("pre_linear must keep the channel count (K4 reads a 128-channel incoming map)")
                wd = w.double()
                n = wt.shape[0]
                blocks = range(0, self.trans_in_dim, n) if level0 else [self.dh_dim]
                wx_sum = sum(wd[:, o:o + n] for o in blocks)
                wd = wd.clone()
                for o in blocks:
                    wd[:, o:o + n] = wd[:, o:o + n] @ wt
                w = wd.float()
                if pre[1] is not None:
                    b = (conv.bias.detach().double() + wx_sum @ pre[1].detach().double()).float()
            ent = (w.to(torch.float16 if form == "fp16" else BF16).contiguous(), b)     # (bf16 weights in the bf16-in-fp16 form as well)
            # Entries may be baked into a live hipGraph (SlotClipRunner, detector._head_cache): an entry of the CURRENT weight version is
            # never evicted (at most form x level0 x pre = a handful per version); entries of older weight versions go - a graph keyed
            # on them is re-captured by its owner anyway (both owners key their graphs on the parameter versions)
            for k in [k for k in cache if k[:4] != key[:4]]:
                del cache[k]
            while len(cache) >= 32:                      # (a caller cycling through many pre_linear tensors on one weight version)
                cache.pop(next(iter(cache)))
            cache[key] = ent
        return ent

    def fuse_level(self, cur, prev_pm, hw, last=False, pre=None):
        """K4 (:171-188). cur [T, 128, H, W] fp32 (the reference's layout) or [T, H*W, 128] 16-bit pixel-major (bf16; fp16 with
        map_dtype "fp16"); prev_pm [T, (H/2)*(W/2), 256] fused map of the coarser level or None (level 0); pre: see _conv_weights.
        Returns the fused map [T, H*W, 256] pixel-major."""
        if pre is not None and self.precision not in ("bf16", "fp16x2"):
            raise NotImplementedError("a folded pre_linear needs the 16-bit or the fp16x2 level fusion (precision 'bf16' / 'fp16x2')")
        if self.precision == "fp32":
            conv = self.conv_trans.conv
            wT = _cached(self, "cwT", [conv.weight], lambda: conv.weight.reshape(self.dh_dim, self.trans_in_dim).t().contiguous())
            return ops.level_fuse_f32(cur.float().contiguous(), prev_pm, wT, conv.bias, hw[0], hw[1])
        if self.precision == "fp16x2":
            conv = self.conv_trans.conv
            planes_in = cur.dim() == 4 and cur.dtype == torch.float16       # [2 (hi, lo), T, HW, 128]: the semantic tower's own rows
            if cur.dim() != 4 or (pre is not None and not planes_in):
                raise NotImplementedError("precision 'fp16x2' takes the incoming maps as [T, 128, H, W] fp32 (NCHW) or, with a folded "
                                          "pre_linear, as fp16 hi + lo pixel-major planes [2, T, H*W, 128]")
            # the level recursion without any 256-wide product (csrc/level_fuse_hl.hip): G^(m)_i = f_i (W_a^m)^T = up(G^(m+1)_{i-1}) +
            # (W_a^m W_b) x_i + W_a^m b. `level` counts from the coarsest; level i produces the orders m = 0 .. (levels_left) its finer
            # levels will ask for: m = 0 the planes of f_i, m >= 1 fp32 only
            srcs = [conv.weight, conv.bias] + ([t for t in pre if t is not None] if pre is not None else [])
            cw = _cached(self, "cw_hl_composed" + ("_pre" if pre is not None else ""), srcs,
                         lambda: ops.level_fuse_hl_composed(conv.weight, conv.bias, self.feat_num_levels, pre=pre))
            if prev_pm is None:
                level, gp = 0, {}
            else:
                level, gp = prev_pm._svps_level + 1, prev_pm._svps_g
            orders = 1 if last else max(1, self.feat_num_levels - level)
            cur_in = cur.contiguous() if planes_in else cur.float().contiguous()
            out, g = None, {}
            for m in range(orders):
                if prev_pm is not None and (m + 1) not in gp:
                    raise RuntimeError(f"level {level} needs G^({m + 1}) of the level below (was that level fused with last=True?)")
            w_hls = [cw["w0"][m] if prev_pm is None else cw["w"][m] for m in range(orders)]
            biases = [cw["b0"][m] if prev_pm is None else cw["b"][m] for m in range(orders)]
            if self.fuse_orders_in_one_launch and orders > 1:
                # every order of the level from ONE launch: the n workgroups of a chunk of tiles share the incoming tile through L2
                # (csrc/level_fuse_hl.hip, level_fuse_hl_multi_kernel); bit-identical to the per-order launches below
                out, f32s = ops.level_fuse_hl_orders(cur_in, None if prev_pm is None else [gp[m + 1] for m in range(orders)], w_hls, biases,
                                                     hw[0], hw[1])
                g = {m: f32s[m] for m in range(1, orders)}
            else:
                for m in range(orders):
                    planes, f32 = ops.level_fuse_hl_g(cur_in, None if prev_pm is None else gp[m + 1], w_hls[m], biases[m], hw[0], hw[1],
                                                      planes=m == 0, f32=m > 0)
                    if m == 0:
                        out = planes
                    else:
                        g[m] = f32
            out._svps_level, out._svps_g = level, g
            return out
        form = self._map_form(cur)
        if prev_pm is not None and form != "fp16":                       # a level follows the encoding of the level below it
            form = "bf16_in_fp16" if prev_pm.dtype == torch.float16 else "bf16"
        wc, bc = self._conv_weights(form, pre, level0=prev_pm is None)
        if cur.dim() == 4:
            cur = cur.float()                                              # NCHW: the reference's fp32 map
        elif cur.dtype != wc.dtype:
            cur = cur.to(wc.dtype)                                         # pixel-major rows: the conv's operand type
        return ops.level_fuse(cur.contiguous(), prev_pm, wc, bc, hw[0], hw[1], bf16_values=form == "bf16_in_fp16")

    def forward_clip(self, feats, init_slots, pos_tabs, hws=None, clip_frames=None, pre_linear=None, pixel_stream=None):
        """Batched clip entry. clip_frames: frames per clip when several clips of equal length are stacked along T
        (T % clip_frames == 0): every kernel then covers all of them in one launch and the temporal slot attention
        stays inside each clip. None = one clip of T frames (the reference's call).
        feats: list over the 4 levels (coarse -> fine) of [T, 128, Hi, Wi] fp32 (NCHW, the reference's
        layout), [T, Hi*Wi, 128] 16-bit pixel-major (16-bit modes) or [2 (hi, lo), T, Hi*Wi, 128] fp16 planes (mode fp16x2) - the
        pixel-major forms need hws = [(Hi, Wi)]; pre_linear = (W_t, b_t): a linear 1x1 map
        in front of the head folded into K4's weights (_conv_weights: the detector's conv_trans - the feats are then ITS input, the
        semantic tower's own output); init_slots [L, 256];
        pos_tabs: per level the separable sine tables (ytab [Hi, 128], xtab [Wi, 128]) of
        ops.pos_embed_sine_tables, or None for no position embedding.
        pixel_stream (mode fp16x2): a second stream for the PIXEL side. Level fusion (K4) and the LayerNorm statistics (K3) of every stage
        depend on the incoming maps and the weights only, never on the slots; the slot chain (self-attention, query side, K1', feed-forward,
        temporal step, towers: ~40 small launches per stage that leave most of the chip idle) is the other dependency chain. With a stream
        given, all of K4 / K3 is issued on it first, each stage's retriever waits for its own statistics, and the streams join at the end -
        inside a captured hipGraph the two chains become parallel branches (clip.SlotClipRunner). Same kernels, same order per tensor:
        bitwise the single-stream result.
        Returns logits [S, T, L, nc], embeds [S, T, L, 256], fused list of [T, Hi*Wi, 256] bf16 (fp32 in exact mode; precision "fp16x2":
        [2, T, Hi*Wi, 256] fp16, the hi and lo planes)."""
        if not feats[0].is_cuda:
            raise RuntimeError("MultiScaleDynamicMaskHead runs on the GPU only; there is no CPU fallback")
        planes_in = feats[0].dim() == 4 and feats[0].dtype == torch.float16
        T = feats[0].shape[1] if planes_in else feats[0].shape[0]
        clips = 1 if clip_frames is None else T // clip_frames
        if clip_frames is not None and clips * clip_frames != T:
            raise ValueError(f"T={T} is not a multiple of clip_frames={clip_frames}")
        # the initial slots of every frame: a broadcast, never written. Cached per (parameter version, T) only when the caller hands
        # over a model PARAMETER (the detector's init_mask_query.weight): (data_ptr, _version) does not identify a temporary - a second
        # temporary with other values can land on the same allocator address at version 0
        if isinstance(init_slots, nn.Parameter):
            slots = _cached(self, f"init_slots_T{T}", [init_slots], lambda: init_slots.float().unsqueeze(0).expand(T, -1, -1).contiguous())
        else:
            slots = init_slots.detach().float().unsqueeze(0).expand(T, -1, -1).contiguous()
        n_stages = sum(self.per_dh_num_heads[:self.feat_num_levels])
        direct = self.precision != "fp32"                # the stages' producers write straight into the stacked results
        out_logits = torch.empty((n_stages, T, init_slots.shape[0], self.num_classes), dtype=torch.float32, device=slots.device) if direct else None
        out_embeds = torch.empty((n_stages, T, init_slots.shape[0], self.dh_dim), dtype=torch.float32, device=slots.device) if direct else None
        all_logits, all_embeds, fused = [], [], []
        prev = None
        stage_idx = 0
        level_hw = [feats[i].shape[-2:] if (feats[i].dim() == 4 and not planes_in) else hws[i] for i in range(self.feat_num_levels)]
        ahead = None
        if pixel_stream is not None and self.precision == "fp16x2":
            # the pixel side of ALL levels on its own stream (see the docstring); every tensor it allocates stays referenced until the streams
            # have joined, so the allocator cannot hand one of its blocks out again while the other stream still reads it
            main = torch.cuda.current_stream(slots.device)
            pixel_stream.wait_stream(main)
            ahead = []
            with torch.cuda.stream(pixel_stream):
                p_ = None
                for i in range(self.feat_num_levels):
                    h, w = level_hw[i]
                    p_ = self.fuse_level(feats[i], p_, (h, w), last=i == self.feat_num_levels - 1, pre=pre_linear)
                    tabs_i = None if pos_tabs is None else pos_tabs[i]
                    per_stage = []
                    for stage in getattr(self, f"head_series_{i}"):
                        aux = stage.inst_interact.stats_hl(p_, (h, w), tabs_i)
                        ev = torch.cuda.Event()
                        ev.record(pixel_stream)
                        per_stage.append((aux, ev))
                    ahead.append((p_, per_stage))
        for i in range(self.feat_num_levels):
            h, w = level_hw[i]
            if ahead is not None:
                f_pm = ahead[i][0]
                for stage, (aux, ev) in zip(getattr(self, f"head_series_{i}"), ahead[i][1]):
                    stage.inst_interact._level_stats = (f_pm, aux, ev)
            else:
                f_pm = self.fuse_level(feats[i], prev, (h, w), last=i == self.feat_num_levels - 1, pre=pre_linear)
            series = getattr(self, f"head_series_{i}")
            mdcs = [stage.inst_interact for stage in series]
            if (self.stats_form == "level" and len(mdcs) == 2 and f_pm.dtype in (BF16, torch.float16) and f_pm.dim() == 3
                    and all(m.precision == "bf16" and m.retriever == "fused" for m in mdcs)):
                # K3'': the LayerNorm statistics of both stages of this level from ONE read of the fused map (csrc/retr_stats2.hip;
                # measured 195 against 2 x 116 us at the finest level); each stage's retriever picks its rows up in forward_fused.
                # A level with a single stage keeps K3'
                tabs_i = None if pos_tabs is None else pos_tabs[i]
                for m, aux in zip(mdcs, ops.retr_stats_level(f_pm, h, w, [m.stats_args(tabs_i) for m in mdcs])):
                    m._level_stats = (f_pm, aux)
            for stage in series:
                enable = stage_idx in self.apply_temporal_query_atten_stages
                logits, slots = stage.forward_pm(slots, f_pm, (h, w), None if pos_tabs is None else pos_tabs[i], enable, clips,
                                                 out_cls=out_logits[stage_idx] if direct else None,
                                                 out_emb=out_embeds[stage_idx] if direct else None)
                slots = slots.detach()
                all_logits.append(logits)
                all_embeds.append(slots)
                stage_idx += 1
            prev = f_pm
            fused.append(f_pm)
        if ahead is not None:
            torch.cuda.current_stream(slots.device).wait_stream(pixel_stream)      # join (every stage has waited for its own event already)
        if direct and all(t.data_ptr() == out_logits[j].data_ptr() for j, t in enumerate(all_logits)) \
                and all(t.data_ptr() == out_embeds[j].data_ptr() for j, t in enumerate(all_embeds)):
            return out_logits, out_embeds, fused
        return torch.stack(all_logits), torch.stack(all_embeds), fused

    def forward(self, features, init_masks, pad_mask, pos=None, query_pos=None, gt_non_void_mask=None):
        """Reference contract (:138-228): features[t][i] [1, 128, Hi, Wi], init_masks[t] [L, 256] (replaced
        in place by their batched [1, L, 256] form like the reference does, :152), pos[t][i] [1, 256, Hi, Wi].
        Returns ([T x [S, 1, L, nc]], [T x [S, 1, L, 256]], [T][4] fused maps [1, 256, Hi, Wi])."""
        assert pad_mask is None and query_pos is None and gt_non_void_mask is None
        T, nlev = len(features), len(features[0])
        assert features[0][0].shape[0] == 1, "batch size 1 per frame (vps_temporal_slots.py:483-484)"
        feats = [torch.cat([features[t][i] for t in range(T)], 0) for i in range(nlev)]
        pos_tabs = [pos_tables_from_map(pos[0][i]) for i in range(nlev)] if pos is not None else None
        init = init_masks[0]
        for t in range(T):
            init_masks[t] = init_masks[t][None]
        logits, embeds, fused = self.forward_clip(feats, init, pos_tabs)
        ret_feats = []
        for t in range(T):
            per = []
            for i in range(nlev):
                _, _, h, w = feats[i].shape
                f = fused[i]
                # precision "fp16x2": the map is two fp16 planes [2 (hi, lo), T, HW, 256] - the frame is their sum (fp32, 22 bits)
                ft = (f[0, t].float() + f[1, t].float()) if f.dim() == 4 else f[t]
                per.append(ft.view(h, w, -1).permute(2, 0, 1).unsqueeze(0))
            ret_feats.append(per)
        if self.return_intermediate:
            return ([logits[:, t:t + 1] for t in range(T)], [embeds[:, t:t + 1] for t in range(T)], ret_feats)
        return [logits[-1, t:t + 1][None] for t in range(T)], [embeds[-1, t:t + 1][None] for t in range(T)], None


def fold_bn_eval(bn):
    """Eval-mode BatchNorm -> (scale, shift) fp32 vectors."""
    scale = bn.weight.float() / torch.sqrt(bn.running_var.float() + bn.eps)
    return scale.contiguous(), (bn.bias.float() - bn.running_mean.float() * scale).contiguous()


def generate_final_outputs(feat_pm, slot_embed, feat_bn, fg_bn, want_argmax=False):
    """K2 wrapper with the semantics of VPS_Temporal_Slots.generate_final_outputs(aux=False)
    (vps_temporal_slots.py:144-160): feat_pm [T, HW, 256] bf16 finest fused map, slot_embed
    [T, L, 256] last-stage embeddings -> mask logits [T, L, HW] fp32 (+ uint8 slot argmax [T, HW])."""
    scale, shift = fold_bn_eval(feat_bn)
    fs, fb = fold_bn_eval(fg_bn)
    if feat_pm.dim() == 4:                                   # precision "fp16x2": fp16 hi + lo planes
        return ops.mask_decode_hl(feat_pm, slot_embed.float().contiguous(), scale, shift, float(fs.item()), float(fb.item()),
                                  want_argmax=want_argmax)
    if feat_pm.dtype == torch.float32:                       # exact mode
        masks = ops.mask_decode_f32(feat_pm, slot_embed.float().contiguous(), scale, shift, float(fs.item()), float(fb.item()))
        return (masks, masks.argmax(dim=1).to(torch.uint8)) if want_argmax else masks
    return ops.mask_decode(feat_pm, slot_embed.float().contiguous(), scale, shift, float(fs.item()), float(fb.item()),
                           want_argmax=want_argmax)
