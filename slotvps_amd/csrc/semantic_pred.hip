// The prediction layer of the semantic head in one kernel:
//     fcn_score = conv1x1( cat( p0, up2(p1), up4(p2), up8(p3) ) )           mmdet/models/panoptic/upsnetFPN.py (forward)
// The framework ran three bilinear upsamplings to the finest level, a 512-channel concatenation and a 1 x 1 convolution: 2.5 ms per
// T = 5 clip, most of it writing and re-reading the 512-channel tensor. Here one thread owns one output pixel: per channel it blends the
// four taps of each coarser level with torch's upsample_bilinear2d arithmetic (align_corners = False, scale 1 / 2^i; -ffp-contract=off)
// and accumulates the K <= 32 class scores in registers, channel by channel in the concatenation's order; the weights sit in LDS as
// [4 C][K] rows (broadcast reads). fp32 throughout. The summation order of the 512 products is this kernel's own (the framework's
// convolution has its own, too): the scores agree with the framework's to fp32 rounding.
#include "common.h"
#include "../../include/slotvps_hip.h"

namespace svps {

constexpr int kSpK = 32;          // class scores per pixel, at most

__device__ __forceinline__ void sp_axis(int dst, float scale, int n_in, int& i0, int& i1, float& l0, float& l1) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;          // area_pixel_compute_source_index, align_corners=False
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    i0 = i0 < n_in - 1 ? i0 : n_in - 1;
    i1 = i0 + 1 < n_in - 1 ? i0 + 1 : n_in - 1;
    l1 = src - (float)i0;
    l0 = 1.f - l1;
}

struct SpArgs {
    const float* p[4];            // level maps [N, C, H >> i, W >> i] fp32 NCHW, finest first
    const float* w;               // [K, 4 C] conv weight
    const float* b;               // [K] or null
    float* out;                   // [N, K, H, W]
    int C, K, H, W;
};

// workgroup = 256 x 4 pixels (1 024 threads sharing one weight tile in LDS: with 256-thread workgroups the 40 - 64 KB tile capped the CU at
// two to four workgroups - two waves per SIMD, a latency-bound kernel: 5.1 ms for VIPER's ten frames and 23 classes)
template <int KU>                 // class scores kept per thread: 20 (Cityscapes' 19), 24 (VIPER's 23) or 32
__global__ __launch_bounds__(1024) void semantic_pred_kernel(SpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float wl[];          // [4 C][KU]
    const int tid = threadIdx.y * 256 + threadIdx.x;
    const int C4 = 4 * a.C;
    for (int e = tid; e < C4 * KU; e += 1024) {
        const int ch = e / KU, k = e - ch * KU;
        wl[e] = k < a.K ? a.w[(size_t)k * C4 + ch] : 0.f;
    }
    __syncthreads();
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, n = blockIdx.z;
    if (x >= a.W || y >= a.H) return;
    float acc[KU];
#pragma unroll
    for (int k = 0; k < KU; ++k) acc[k] = 0.f;
    // level 0: the pixel itself
    {
        const size_t hw = (size_t)a.H * a.W;
        const float* src = a.p[0] + (size_t)n * a.C * hw + (size_t)y * a.W + x;
        for (int c = 0; c < a.C; ++c) {
            const float v = src[(size_t)c * hw];
            const float* wr = wl + c * KU;
#pragma unroll
            for (int k = 0; k < KU; ++k) acc[k] += wr[k] * v;
        }
    }
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        const int h = a.H >> i, w = a.W >> i;
        const float scale = 1.f / (float)(1 << i);
        int y0, y1, x0, x1;
        float hy0, hy1, wx0, wx1;
        sp_axis(y, scale, h, y0, y1, hy0, hy1);
        sp_axis(x, scale, w, x0, x1, wx0, wx1);
        const size_t hw = (size_t)h * w;
        const float* src = a.p[i] + (size_t)n * a.C * hw;
        const int o00 = y0 * w + x0, o01 = y0 * w + x1, o10 = y1 * w + x0, o11 = y1 * w + x1;
        for (int c = 0; c < a.C; ++c) {
            const float* m = src + (size_t)c * hw;
            const float v = hy0 * (wx0 * m[o00] + wx1 * m[o01]) + hy1 * (wx0 * m[o10] + wx1 * m[o11]);
            const float* wr = wl + (i * a.C + c) * KU;
#pragma unroll
            for (int k = 0; k < KU; ++k) acc[k] += wr[k] * v;
        }
    }
    const size_t hw = (size_t)a.H * a.W;
    float* dst = a.out + (size_t)n * a.K * hw + (size_t)y * a.W + x;
#pragma unroll
    for (int k = 0; k < KU; ++k)
        if (k < a.K) dst[(size_t)k * hw] = acc[k] + (a.b ? a.b[k] : 0.f);
}

}  // namespace svps

extern "C" int svps_semantic_pred_fwd(const float* p0, const float* p1, const float* p2, const float* p3, const float* weight,
                                      const float* bias, float* out, int N, int C, int K, int H, int W, void* stream_) {
    if (!p0 || !p1 || !p2 || !p3 || !weight || !out) return SVPS_ERR_BAD_ARG;
    if (N <= 0 || C <= 0 || K <= 0 || K > svps::kSpK || H <= 0 || W <= 0 || (H & 7) || (W & 7)) return SVPS_ERR_BAD_SHAPE;
    const int ku = K <= 20 ? 20 : (K <= 24 ? 24 : svps::kSpK);
    const size_t lds = (size_t)4 * C * ku * sizeof(float);
    if (lds > 160 * 1024) return SVPS_ERR_BAD_SHAPE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    svps::SpArgs a{{p0, p1, p2, p3}, weight, bias, out, C, K, H, W};
    const dim3 grid((W + 255) / 256, (H + 3) / 4, N);
    if (ku == 20) {
        static SvpsLdsAttr attr;
        if (lds > 48 * 1024)
            if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(svps::semantic_pred_kernel<20>), (int)lds); e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(svps::semantic_pred_kernel<20>, grid, dim3(256, 4), lds, stream, a);
    } else if (ku == 24) {                                    // VIPER's 23 classes
        static SvpsLdsAttr attr;
        if (lds > 48 * 1024)
            if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(svps::semantic_pred_kernel<24>), (int)lds); e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(svps::semantic_pred_kernel<24>, grid, dim3(256, 4), lds, stream, a);
    } else {
        static SvpsLdsAttr attr;
        if (lds > 48 * 1024)
            if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(svps::semantic_pred_kernel<32>), (int)lds); e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(svps::semantic_pred_kernel<32>, grid, dim3(256, 4), lds, stream, a);
    }
    return (int)hipGetLastError();
}
