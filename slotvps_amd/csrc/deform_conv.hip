// K7 - deformable convolution forward (DCNv1) for gfx950.
//
// Replaces deform_conv_forward_cuda of the reference (mmdet/ops/dcn/src/deform_conv_cuda.cpp:152-258 with the
// sampling kernel deformable_im2col_gpu_kernel, deform_conv_cuda_kernel.cu:190-241, bilinear rule :82-114) -
// the only native op reached at inference (UPSNetFPN, mmdet/models/panoptic/upsnetFPN.py:36-49: 3 layers x 4
// levels per frame). Same decomposition as the reference: deformable im2col, then a GEMM with the flattened
// weight (done by the caller).
//
// Layout: input pixel-major (NHWC) so that a wavefront reads 64 consecutive channels of one tap (256 B) and
// writes 64 x 9 consecutive column entries; offsets in the reference's layout [N, dg*2*kh*kw, Ho, Wo]
// (channel 2*(i*kw+j) = dy, +1 = dx). Columns [N, Ho*Wo, C*kh*kw] with column index c*kh*kw + i*kw + j, i.e.
// weight.view(O, -1) is the matching GEMM operand.
// A sample outside (-1, H) x (-1, W) is zero; corners outside the image contribute zero.
#include <hip/hip_runtime.h>

#include "../../include/slotvps_hip.h"

namespace svps {

__global__ __launch_bounds__(256) void deform_im2col_kernel(const float* __restrict__ x,       // [N, H, W, C]
                                                            const float* __restrict__ offset,  // [N, dg*2*kh*kw, Ho, Wo]
                                                            float* __restrict__ cols,          // [N, Ho*Wo, C*kh*kw]
                                                            int N, int C, int H, int W, int kh, int kw, int pad_h,
                                                            int pad_w, int stride_h, int stride_w, int dil_h, int dil_w,
                                                            int dg, int Ho, int Wo) {
    const size_t total = (size_t)N * Ho * Wo * C;
    const int ktaps = kh * kw;
    const int cpg = C / dg;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = (int)(idx % C);
        const size_t p = idx / C;                      // n * Ho*Wo + pixel
        const int wo = (int)(p % Wo);
        const int ho = (int)((p / Wo) % Ho);
        const int n = (int)(p / ((size_t)Wo * Ho));
        const int g = c / cpg;
        const float* xin = x + (size_t)n * H * W * C + c;
        const float* off = offset + ((size_t)n * dg + g) * 2 * ktaps * Ho * Wo + (size_t)ho * Wo + wo;
        float* out = cols + (p * C + c) * ktaps;
        const int h_in = ho * stride_h - pad_h, w_in = wo * stride_w - pad_w;
        for (int i = 0; i < kh; ++i)
            for (int j = 0; j < kw; ++j) {
                const int t = i * kw + j;
                const float oh = off[(size_t)(2 * t) * Ho * Wo], ow = off[(size_t)(2 * t + 1) * Ho * Wo];
                const float hf = (float)(h_in + i * dil_h) + oh, wf = (float)(w_in + j * dil_w) + ow;
                float val = 0.f;
                if (hf > -1.f && wf > -1.f && hf < (float)H && wf < (float)W) {
                    const int hl = (int)floorf(hf), wl = (int)floorf(wf);
                    const int hh = hl + 1, wh = wl + 1;
                    const float lh = hf - (float)hl, lw = wf - (float)wl;
                    const float uh = 1.f - lh, uw = 1.f - lw;
                    const float v1 = (hl >= 0 && wl >= 0) ? xin[((size_t)hl * W + wl) * C] : 0.f;
                    const float v2 = (hl >= 0 && wh <= W - 1) ? xin[((size_t)hl * W + wh) * C] : 0.f;
                    const float v3 = (hh <= H - 1 && wl >= 0) ? xin[((size_t)hh * W + wl) * C] : 0.f;
                    const float v4 = (hh <= H - 1 && wh <= W - 1) ? xin[((size_t)hh * W + wh) * C] : 0.f;
                    val = uh * uw * v1 + uh * lw * v2 + lh * uw * v3 + lh * lw * v4;
                }
                out[t] = val;
            }
    }
}

// bf16 operand path: the columns are matrix-core operands, so they are written as bf16 in TAP-MAJOR order
// [N * Ho*Wo, kh*kw, C] (the weight is flattened to [O, kh*kw, C] to match); a thread handles eight consecutive
// channels of one (pixel, tap): four 16-byte neighbour loads from the pixel-major bf16 input, the bilinear blend in
// fp32 (same expression and validity rules as above), one 16-byte store - reads and writes coalesce along channels
// (the fp32 kernel above writes every thread's nine taps 36 B apart).
typedef __attribute__((ext_vector_type(8))) __bf16 dc_bf16x8;

__global__ __launch_bounds__(256) void deform_im2col_bf16_kernel(const __bf16* __restrict__ x,       // [N, H, W, C]
                                                                 const float* __restrict__ offset,  // [N, dg*2*kh*kw, Ho, Wo]
                                                                 __bf16* __restrict__ cols,         // [N, Ho*Wo, kh*kw, C]
                                                                 int N, int C, int H, int W, int kh, int kw, int pad_h,
                                                                 int pad_w, int stride_h, int stride_w, int dil_h,
                                                                 int dil_w, int dg, int Ho, int Wo) {
    const int c8 = C >> 3, ktaps = kh * kw, cpg = C / dg;
    const size_t total = (size_t)N * Ho * Wo * ktaps * c8;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int co = (int)(idx % c8);
        const size_t pt = idx / c8;                    // (n * Ho*Wo + pixel) * ktaps + tap
        const int t = (int)(pt % ktaps);
        const size_t p = pt / ktaps;
        const int wo = (int)(p % Wo);
        const int ho = (int)((p / Wo) % Ho);
        const int n = (int)(p / ((size_t)Wo * Ho));
        const int i = t / kw, j = t - i * kw;
        const int g = (8 * co) / cpg;
        const float* off = offset + (((size_t)n * dg + g) * 2 * ktaps + 2 * t) * Ho * Wo + (size_t)ho * Wo + wo;
        const float hf = (float)(ho * stride_h - pad_h + i * dil_h) + off[0];
        const float wf = (float)(wo * stride_w - pad_w + j * dil_w) + off[(size_t)Ho * Wo];
        dc_bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)0.f;
        if (hf > -1.f && wf > -1.f && hf < (float)H && wf < (float)W) {
            const int hl = (int)floorf(hf), wl = (int)floorf(wf);
            const int hh = hl + 1, wh = wl + 1;
            const float lh = hf - (float)hl, lw = wf - (float)wl;
            const float uh = 1.f - lh, uw = 1.f - lw;
            const __bf16* xin = x + (size_t)n * H * W * C + 8 * co;
            dc_bf16x8 v1 = o, v2 = o, v3 = o, v4 = o;
            if (hl >= 0 && wl >= 0) v1 = *reinterpret_cast<const dc_bf16x8*>(xin + ((size_t)hl * W + wl) * C);
            if (hl >= 0 && wh <= W - 1) v2 = *reinterpret_cast<const dc_bf16x8*>(xin + ((size_t)hl * W + wh) * C);
            if (hh <= H - 1 && wl >= 0) v3 = *reinterpret_cast<const dc_bf16x8*>(xin + ((size_t)hh * W + wl) * C);
            if (hh <= H - 1 && wh <= W - 1) v4 = *reinterpret_cast<const dc_bf16x8*>(xin + ((size_t)hh * W + wh) * C);
            const float w1 = uh * uw, w2 = uh * lw, w3 = lh * uw, w4 = lh * lw;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                o[e] = (__bf16)(w1 * (float)v1[e] + w2 * (float)v2[e] + w3 * (float)v3[e] + w4 * (float)v4[e]);
        }
        *reinterpret_cast<dc_bf16x8*>(cols + pt * C + 8 * co) = o;
    }
}

}  // namespace svps

extern "C" int svps_deform_im2col_bf16(const void* x_nhwc, const float* offset, void* cols, int N, int C, int H, int W,
                                       int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h,
                                       int dil_w, int deformable_groups, int Ho, int Wo, void* stream_) {
    if (!x_nhwc || !offset || !cols) return SVPS_ERR_BAD_ARG;
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || deformable_groups <= 0 ||
        C % deformable_groups || (C / deformable_groups) % 8 || Ho <= 0 || Wo <= 0)
        return SVPS_ERR_BAD_SHAPE;
    if (Ho != (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1 ||
        Wo != (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1)
        return SVPS_ERR_BAD_SHAPE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const size_t total = (size_t)N * Ho * Wo * kh * kw * (C / 8);
    size_t blocks = (total + 255) / 256;
    if (blocks > 262144) blocks = 262144;
    svps_prof_mark(SVPS_KERNEL_DEFORM_CONV, 0, stream);
    hipLaunchKernelGGL(svps::deform_im2col_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                       static_cast<const __bf16*>(x_nhwc), offset, static_cast<__bf16*>(cols), N, C, H, W, kh, kw, pad_h,
                       pad_w, stride_h, stride_w, dil_h, dil_w, deformable_groups, Ho, Wo);
    svps_prof_mark(SVPS_KERNEL_DEFORM_CONV, 1, stream);
    return (int)hipGetLastError();
}

extern "C" int svps_deform_im2col(const float* x_nhwc, const float* offset, float* cols, int N, int C, int H, int W,
                                  int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h,
                                  int dil_w, int deformable_groups, int Ho, int Wo, void* stream_) {
    if (!x_nhwc || !offset || !cols) return SVPS_ERR_BAD_ARG;
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || deformable_groups <= 0 ||
        C % deformable_groups || Ho <= 0 || Wo <= 0)
        return SVPS_ERR_BAD_SHAPE;
    if (Ho != (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1 ||
        Wo != (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1)
        return SVPS_ERR_BAD_SHAPE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const size_t total = (size_t)N * Ho * Wo * C;
    size_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    svps_prof_mark(SVPS_KERNEL_DEFORM_CONV, 0, stream);
    hipLaunchKernelGGL(svps::deform_im2col_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x_nhwc, offset, cols, N,
                       C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, deformable_groups, Ho, Wo);
    svps_prof_mark(SVPS_KERNEL_DEFORM_CONV, 1, stream);
    return (int)hipGetLastError();
}
