"""Deformable convolution modules on the HIP path: mirrors of the reference's DeformConv
(mmdet/ops/dcn/deform_conv.py:217-263) and DeformConvWithOffset (mmdet/models/utils/deform_conv_with_offset.py)
with the same parameter names (`weight`; `conv_offset.{weight,bias}`, `conv.weight`). Forward: K7' (one kernel, no
column buffer, split-bf16 matrix-core products with fp32 accumulation: csrc/deform_conv_fused.hip) for the shapes of the
semantic tower (3 x 3, one deformable group, C % 64 == 0, 128 or 256 output channels); K7 deformable im2col + one GEMM
(the reference's own decomposition, deform_conv_cuda.cpp:152-258) otherwise."""
import ctypes
import math

import torch
from torch import nn
from torch.nn.modules.utils import _pair

from . import _lib, ops


def pack_weight_fragments(weight):
    """[O, C, 3, 3] fp32 -> bf16 [O/32, 9C/16, 2, 64, 8]: hi / lo halves in MFMA A-fragment order, k = tap * C + c
    (the layout svps_deform_conv_fused_fwd streams: one 1-KiB wave instruction per fragment)."""
    O, C, kh, kw = weight.shape
    wt = weight.detach().float().permute(0, 2, 3, 1).reshape(O, kh * kw * C)          # [O, k = t*C + c]
    hi = wt.to(torch.bfloat16)
    lo = (wt - hi.float()).to(torch.bfloat16)
    ks = kh * kw * C // 16

    def frag(m):                                   # (ob, r, ks, h, j) -> (ob, ks, h, r, j): lane = 32 h + r
        return m.view(O // 32, 32, ks, 2, 8).permute(0, 2, 3, 1, 4).reshape(O // 32, ks, 64, 8)
    return torch.stack([frag(hi), frag(lo)], dim=2).contiguous()


def fused_applicable(x, weight, stride, padding, dilation, groups, deformable_groups):
    O, C, kh, kw = weight.shape
    return (groups == 1 and deformable_groups == 1 and kh == 3 and kw == 3 and C % 64 == 0 and O in (128, 256)
            and _pair(stride)[0] == _pair(stride)[1] and _pair(padding)[0] == _pair(padding)[1]
            and _pair(dilation)[0] == _pair(dilation)[1] and x.shape[2] * x.shape[3] * C < 2 ** 31)


def deform_conv_fused_pm(x_nhwc, offset, wpack, O, stride=1, padding=0, dilation=1, gn_stats=False):
    """K7' on pixel-major tensors: x_nhwc [N, H, W, C] fp32 contiguous, offset [N, 18, Ho, Wo] fp32 -> [N, Ho*Wo, O] fp32.
    gn_stats: also return (partial [N, chunks, 2, O], chunks) - the per-channel sums and sums of squares of the result, written by the
    kernel's epilogue, for ops.group_norm_relu_pm(stats=...)."""
    if not x_nhwc.is_cuda:
        raise RuntimeError("deform_conv runs on the GPU only; there is no CPU fallback")
    lib = _lib.load()
    s, p_, d = _pair(stride)[0], _pair(padding)[0], _pair(dilation)[0]
    N, H, W, C = x_nhwc.shape
    Ho = (H + 2 * p_ - (d * 2 + 1)) // s + 1
    Wo = (W + 2 * p_ - (d * 2 + 1)) // s + 1
    offset = offset.float().contiguous()
    if offset.shape != (N, 18, Ho, Wo) or not x_nhwc.is_contiguous() or x_nhwc.dtype != torch.float32:
        raise ValueError("deform_conv_fused_pm: x [N, H, W, C] fp32 contiguous and offset [N, 18, Ho, Wo] expected")
    out = torch.empty((N, Ho * Wo, O), dtype=torch.float32, device=x_nhwc.device)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
    chunks = lib.svps_deform_conv_fused_stats_chunks(N, O, Ho, Wo) if gn_stats else 0
    part = torch.empty((N, chunks, 2, O), dtype=torch.float32, device=x_nhwc.device) if gn_stats else None
    with ops._on(x_nhwc, offset, wpack, out, part) as ctx:
        rc = lib.svps_deform_conv_fused_stats_fwd(p(x_nhwc), p(offset), p(wpack), p(out), p(part), N, C, H, W, O, 3, 3, p_, s, d, Ho, Wo,
                                                  ctx.stream)
    _lib.check(rc, "svps_deform_conv_fused_stats_fwd")
    return (out, (part, chunks)) if gn_stats else out


def deform_conv_fused(x, offset, wpack, O, stride=1, padding=0, dilation=1):
    """K7': x [N, C, H, W] fp32, offset [N, 18, Ho, Wo], wpack = pack_weight_fragments(weight) -> [N, O, Ho, Wo] fp32."""
    if not x.is_cuda:
        raise RuntimeError("deform_conv runs on the GPU only; there is no CPU fallback")
    lib = _lib.load()
    s, p_, d = _pair(stride)[0], _pair(padding)[0], _pair(dilation)[0]
    N, C, H, W = x.shape
    Ho = (H + 2 * p_ - (d * 2 + 1)) // s + 1
    Wo = (W + 2 * p_ - (d * 2 + 1)) // s + 1
    offset = offset.float().contiguous()
    if offset.shape != (N, 18, Ho, Wo):
        raise ValueError(f"offset shape {tuple(offset.shape)}")
    x_nhwc = x.float().permute(0, 2, 3, 1).contiguous()                 # free for channels_last inputs
    out = torch.empty((N, Ho * Wo, O), dtype=torch.float32, device=x.device)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    with ops._on(x_nhwc, offset, wpack, out) as ctx:
        rc = lib.svps_deform_conv_fused_fwd(p(x_nhwc), p(offset), p(wpack), p(out), N, C, H, W, O, 3, 3, p_, s, d, Ho, Wo, ctx.stream)
    _lib.check(rc, "svps_deform_conv_fused_fwd")
    return out.view(N, Ho, Wo, O).permute(0, 3, 1, 2)                    # NCHW view over channels_last memory


def deform_conv(x, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1, bf16_operands=False,
                weight_taps=None):
    """x [N, C, H, W] fp32 (any memory format), offset [N, dg*2*kh*kw, Ho, Wo], weight [O, C, kh, kw] -> [N, O, Ho, Wo].
    bf16_operands: sampled columns and weights as bf16 matrix-core operands with fp32 accumulation (the storage policy of
    the feature path; ~6x faster than the fp32 columns + fp32 GEMM); weight_taps = cached
    weight.permute(0, 2, 3, 1).reshape(O, -1) in bf16."""
    if not x.is_cuda:
        raise RuntimeError("deform_conv runs on the GPU only (the reference has no CPU path either, "
                           "mmdet/ops/dcn/deform_conv.py:44-45); there is no CPU fallback")
    if groups != 1:
        raise NotImplementedError("groups != 1 is not used by the Slot-VPS configs")
    lib = _lib.load()
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    dh, dw = _pair(dilation)
    N, C, H, W = x.shape
    O, _, kh, kw = weight.shape
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    offset = offset.float().contiguous()
    if offset.shape != (N, deformable_groups * 2 * kh * kw, Ho, Wo):
        raise ValueError(f"offset shape {tuple(offset.shape)}")
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    if bf16_operands and (C // deformable_groups) % 8 == 0:
        x_nhwc = torch.empty((N, H, W, C), dtype=torch.bfloat16, device=x.device)
        x_nhwc.copy_(x.permute(0, 2, 3, 1))                              # cast + transpose in one pass
        cols = torch.empty((N * Ho * Wo, kh * kw * C), dtype=torch.bfloat16, device=x.device)
        with ops._on(x_nhwc, offset, cols) as ctx:
            rc = lib.svps_deform_im2col_bf16(p(x_nhwc), p(offset), p(cols), N, C, H, W, kh, kw, ph, pw, sh, sw, dh, dw,
                                             deformable_groups, Ho, Wo, ctx.stream)
        _lib.check(rc, "svps_deform_im2col_bf16")
        if weight_taps is None:
            weight_taps = weight.permute(0, 2, 3, 1).reshape(O, -1).to(torch.bfloat16).contiguous()
        out = torch.mm(cols, weight_taps.t(), out_dtype=torch.float32)   # [N*Ho*Wo, O], fp32 accumulate and output
        return out.view(N, Ho, Wo, O).permute(0, 3, 1, 2)
    x_nhwc = x.float().permute(0, 2, 3, 1).contiguous()                 # free for channels_last inputs
    cols = torch.empty((N, Ho * Wo, C * kh * kw), dtype=torch.float32, device=x.device)
    with ops._on(x_nhwc, offset, cols) as ctx:
        rc = lib.svps_deform_im2col(p(x_nhwc), p(offset), p(cols), N, C, H, W, kh, kw, ph, pw, sh, sw, dh, dw,
                                    deformable_groups, Ho, Wo, ctx.stream)
    _lib.check(rc, "svps_deform_im2col")
    out = cols @ weight.reshape(O, -1).t().float()                       # [N, Ho*Wo, O]
    return out.view(N, Ho, Wo, O).permute(0, 3, 1, 2)                    # NCHW view over channels_last memory


class DeformConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=False):
        super().__init__()
        assert not bias
        assert in_channels % groups == 0 and out_channels % groups == 0
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride, self.padding, self.dilation = _pair(stride), _pair(padding), _pair(dilation)
        self.groups, self.deformable_groups = groups, deformable_groups
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *self.kernel_size))
        # False (default): fp32 columns and fp32 GEMM, like the reference's op. True: bf16 columns / weights on the matrix
        # cores with fp32 accumulation - switched together with the detector's `trunk_bf16` (detector.py), never silently
        self.bf16_operands = False
        self.fused = True                    # K7' (no column buffer) where its shape constraints hold
        self._wt = None
        self._wp = None
        self.reset_parameters()

    def reset_parameters(self):
        n = self.in_channels
        for k in self.kernel_size:
            n *= k
        stdv = 1. / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)

    def _weight_taps(self):
        key = (self.weight._version, self.weight.data_ptr())
        if self._wt is None or self._wt[0] != key:
            O = self.weight.shape[0]
            self._wt = (key, self.weight.detach().permute(0, 2, 3, 1).reshape(O, -1).to(torch.bfloat16).contiguous())
        return self._wt[1]

    def _weight_pack(self):
        key = (self.weight._version, self.weight.data_ptr())
        if self._wp is None or self._wp[0] != key:
            self._wp = (key, pack_weight_fragments(self.weight))
        return self._wp[1]

    def forward(self, x, offset):
        if (self.fused and not self.bf16_operands and x.is_cuda
                and fused_applicable(x, self.weight, self.stride, self.padding, self.dilation, self.groups, self.deformable_groups)):
            return deform_conv_fused(x, offset, self._weight_pack(), self.weight.shape[0], self.stride, self.padding, self.dilation)
        return deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups,
                           self.deformable_groups, self.bf16_operands, self._weight_taps() if self.bf16_operands else None)


class DeformConvWithOffset(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=True):
        super().__init__()
        self.conv_offset = nn.Conv2d(in_channels, kernel_size * kernel_size * 2 * deformable_groups, kernel_size=3,
                                     stride=1, padding=1)
        self.conv_offset.weight.data.zero_()
        self.conv_offset.bias.data.zero_()
        self.conv = DeformConv(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                               dilation=dilation, groups=groups, deformable_groups=deformable_groups, bias=False)

    def forward(self, x):
        return self.conv(x, self.conv_offset(x))
