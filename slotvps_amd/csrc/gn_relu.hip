// GroupNorm + ReLU on pixel-major (NHWC) fp32 activations, for the semantic tower.
//
// UPSNetFPN's tower (mmdet/models/panoptic/upsnetFPN.py:36-49) is three [deformable conv 3x3 -> GroupNorm(32) -> ReLU] per pyramid
// level. K7' (deform_conv_fused.hip) reads and writes pixel-major fp32; the framework's GroupNorm wants NCHW, so every layer paid a
// pixel-major -> NCHW copy inside GroupNorm, the moments kernel, the normalisation kernels and an NCHW -> pixel-major copy in front
// of the next K7' (profiles: 8.5 of the tower's 24 ms per T = 5 clip). Here:
//   gn_stats_kernel     per (frame, chunk of pixels): per-CHANNEL sums and sums of squares, one float4 (four channels) per thread
//   gn_finalize_kernel  per frame: chunks -> channels -> groups in float64; per-channel scale a = rstd_g gamma_c and
//                       shift b = beta_c - mean_g rstd_g gamma_c
//   gn_apply_kernel     y = max(a x + b, 0) pixel-major (the next K7' reads it) AND, if asked, NCHW (the offset-producing 3x3 conv of
//                       the next layer and the tower's consumers are framework convolutions): 32-pixel tiles transposed through LDS,
//                       128-byte rows per channel
// fp32 arithmetic, float64 only where sums over ~1e6 elements are combined. Any C with C % 4 == 0 and C % groups == 0.
#include "common.h"
#include "../../include/slotvps_hip.h"

namespace svps {

__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, float* __restrict__ partial, int HW, int C,
                                                       int px_per_chunk) {
    __shared__ float red[2 * 256 * 4];
    const int n = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int cols = C >> 2;                                   // float4 columns of a pixel row
    const int rows = 256 / cols;                               // pixels per sweep of the workgroup (cols <= 256: C <= 1024)
    const int pr = tid / cols, cc = tid - pr * cols;
    const int p0 = chunk * px_per_chunk;
    int p1 = p0 + px_per_chunk;
    p1 = p1 < HW ? p1 : HW;
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, ss = {0.f, 0.f, 0.f, 0.f};
    if (pr < rows) {
        const float* base = x + ((size_t)n * HW) * C + 4 * cc;
        for (int p = p0 + pr; p < p1; p += rows) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)p * C);
#pragma unroll
            for (int j = 0; j < 4; ++j) { s[j] += v[j]; ss[j] += v[j] * v[j]; }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[tid * 4 + j] = s[j]; red[1024 + tid * 4 + j] = ss[j]; }
    __syncthreads();
    if (tid < cols) {                                          // column tid: sum over the pixel rows of the sweep
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < rows; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) { a[j] += red[(q * cols + tid) * 4 + j]; b[j] += red[1024 + (q * cols + tid) * 4 + j]; }
        float* dst = partial + (((size_t)n * gridDim.x + chunk) * 2) * C + 4 * tid;
        *reinterpret_cast<f32x4*>(dst) = a;
        *reinterpret_cast<f32x4*>(dst + C) = b;
    }
}

__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partial, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float2* __restrict__ ab, int HW, int C,
                                                          int groups, int chunks, float eps) {
    __shared__ double cs[1024], css[1024];
    __shared__ double gm[256], gr[256];
    const int n = blockIdx.x, tid = threadIdx.x;
    for (int c = tid; c < C; c += 256) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < chunks; ++k) {
            const float* src = partial + (((size_t)n * chunks + k) * 2) * C;
            a += (double)src[c];
            b += (double)src[C + c];
        }
        cs[c] = a;
        css[c] = b;
    }
    __syncthreads();
    const int cpg = C / groups;
    for (int g = tid; g < groups; g += 256) {
        double a = 0.0, b = 0.0;
        for (int j = 0; j < cpg; ++j) { a += cs[g * cpg + j]; b += css[g * cpg + j]; }
        const double cnt = (double)HW * cpg;
        const double mean = a / cnt;
        double var = b / cnt - mean * mean;
        var = var > 0.0 ? var : 0.0;
        gm[g] = mean;
        gr[g] = 1.0 / sqrt(var + (double)eps);
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const int g = c / cpg;
        const double sc = gr[g] * (double)gamma[c];
        ab[(size_t)n * C + c] = make_float2((float)sc, (float)((double)beta[c] - gm[g] * sc));
    }
}

// partial rows of a producer kernel ([N, chunks_in, 2, C]: one per 128- or 64-pixel tile, thousands per frame) -> [N, R, 2, C]: workgroup
// (r, n) sums its slice of the rows in float64 (gn_finalize_kernel is one workgroup per frame: it would walk them serially)
__global__ __launch_bounds__(256) void gn_reduce_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int chunks_in, int R) {
    const int n = blockIdx.y, r = blockIdx.x;
    const int per = (chunks_in + R - 1) / R;
    const int k0 = r * per;
    int k1 = k0 + per;
    k1 = k1 < chunks_in ? k1 : chunks_in;
    for (int c = threadIdx.x; c < 2 * C; c += 256) {
        double a = 0.0;
        for (int k = k0; k < k1; ++k) a += (double)in[(((size_t)n * chunks_in + k) * 2) * C + c];
        out[(((size_t)n * R + r) * 2) * C + c] = (float)a;
    }
}

// tile = 32 consecutive pixels x C channels; LDS [32][C + 1] floats for the NCHW copy
// y16 (nullable): the same values as 16-bit pixel-major rows - bf16 (f16 == 0), fp16 (f16 == 1, saturating at +-65 504) or TWO fp16 planes
// hi + lo (f16 == 2: [2, N, HW, C], hi = fp16(x), lo = fp16(x - hi), the operand form of level_fuse_hl.hip: 22 bits of the fp32 value) - the
// form in which the tower's LAST layer hands its output to K4 (conv_trans folded into K4's weights); y (nullable) the fp32 rows the next K7' reads
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const float2* __restrict__ ab, float* __restrict__ y,
                                                       float* __restrict__ y_nchw, void* __restrict__ y16, int f16, int HW, int C,
                                                       int tiles_per_wg) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int n = blockIdx.y, tid = threadIdx.x;
    const int cols = C >> 2, rows = 256 / cols;
    const int pr = tid / cols, cc = tid - pr * cols;
    const int tiles = (HW + 31) >> 5;
    const int t0 = blockIdx.x * tiles_per_wg;
    int t1 = t0 + tiles_per_wg;
    t1 = t1 < tiles ? t1 : tiles;
    f32x4 a4 = {0.f, 0.f, 0.f, 0.f}, b4 = {0.f, 0.f, 0.f, 0.f};
    if (pr < rows) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float2 v = ab[(size_t)n * C + 4 * cc + j];
            a4[j] = v.x;
            b4[j] = v.y;
        }
    }
    const int ldc = C + 1;
    for (int t = t0; t < t1; ++t) {
        const int p0 = t << 5;
        if (pr < rows) {
            for (int q = pr; q < 32; q += rows) {
                const int p = p0 + q;
                if (p < HW) {
                    const size_t off = ((size_t)n * HW + p) * C + 4 * cc;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(x + off);
                    f32x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = fmaxf(fmaf(a4[j], v[j], b4[j]), 0.f);
                    if (y) *reinterpret_cast<f32x4*>(y + off) = o;
                    if (y16) {
                        uint2 pk;
                        if (f16 == 2) {
                            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                            h4 qh, ql;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                float v = fminf(o[j], 65504.f);              // (o >= 0 behind the ReLU)
                                asm volatile("" : "+v"(v));                  // ONE fp32 value for both halves (level_fuse_hl.hip, hl_split)
                                qh[j] = (_Float16)v;
                                ql[j] = (_Float16)(v - (float)qh[j]);
                            }
                            pk = __builtin_bit_cast(uint2, qh);
                            *reinterpret_cast<uint2*>(static_cast<char*>(y16) + ((size_t)gridDim.y * HW * C + off) * 2) = __builtin_bit_cast(uint2, ql);
                        } else if (f16) {
                            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                            h4 q;
#pragma unroll
                            for (int j = 0; j < 4; ++j) q[j] = (_Float16)fminf(o[j], 65504.f);
                            pk = __builtin_bit_cast(uint2, q);
                        } else {
                            typedef __bf16 b4 __attribute__((ext_vector_type(4)));
                            b4 q;
#pragma unroll
                            for (int j = 0; j < 4; ++j) q[j] = (__bf16)o[j];
                            pk = __builtin_bit_cast(uint2, q);
                        }
                        *reinterpret_cast<uint2*>(static_cast<char*>(y16) + off * 2) = pk;
                    }
                    if (y_nchw) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) tile[q * ldc + 4 * cc + j] = o[j];
                    }
                }
            }
        }
        if (y_nchw) {
            __syncthreads();
            // thread -> (channel, pixel quad): 8 quads per channel row of the tile
            for (int e = tid; e < C * 8; e += 256) {
                const int c = e >> 3, q4 = (e & 7) * 4;
                const int p = p0 + q4;
                float* dst = y_nchw + ((size_t)n * C + c) * HW + p;
                if (p + 3 < HW && ((HW & 3) == 0)) {
                    const f32x4 o = {tile[(q4 + 0) * ldc + c], tile[(q4 + 1) * ldc + c], tile[(q4 + 2) * ldc + c], tile[(q4 + 3) * ldc + c]};
                    *reinterpret_cast<f32x4*>(dst) = o;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (p + j < HW) dst[j] = tile[(q4 + j) * ldc + c];
                }
            }
            __syncthreads();
        }
    }
}

// [N, C, HW] fp32 (NCHW) -> [N, HW, C] fp32 pixel-major: the layout copy in front of the tower's first K7' (the framework's
// permute + contiguous ran at a third of the copy rate). 32-pixel tiles through LDS, 128-byte reads per channel row, whole pixel rows out
__global__ __launch_bounds__(256) void nchw_to_pm_kernel(const float* __restrict__ x, float* __restrict__ y, int HW, int C, int tiles_per_wg) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int n = blockIdx.y, tid = threadIdx.x;
    const int ldc = C + 1;
    const int tiles = (HW + 31) >> 5;
    const int t0 = blockIdx.x * tiles_per_wg;
    int t1 = t0 + tiles_per_wg;
    t1 = t1 < tiles ? t1 : tiles;
    const int cols = C >> 2, rows = 256 / cols;
    const int pr = tid / cols, cc = tid - pr * cols;
    for (int t = t0; t < t1; ++t) {
        const int p0 = t << 5;
        for (int e = tid; e < C * 8; e += 256) {
            const int c = e >> 3, q4 = (e & 7) * 4;
            const int p = p0 + q4;
            const float* src = x + ((size_t)n * C + c) * HW + p;
            if (p + 3 < HW && ((HW & 3) == 0)) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
                for (int j = 0; j < 4; ++j) tile[(q4 + j) * ldc + c] = v[j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) tile[(q4 + j) * ldc + c] = p + j < HW ? src[j] : 0.f;
            }
        }
        __syncthreads();
        if (pr < rows) {
            for (int q = pr; q < 32; q += rows) {
                const int p = p0 + q;
                if (p < HW) {
                    const f32x4 o = {tile[q * ldc + 4 * cc], tile[q * ldc + 4 * cc + 1], tile[q * ldc + 4 * cc + 2], tile[q * ldc + 4 * cc + 3]};
                    *reinterpret_cast<f32x4*>(y + ((size_t)n * HW + p) * C + 4 * cc) = o;
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace svps

extern "C" int svps_nchw_to_pixel_major(const float* x, float* y, int N, int C, int HW, void* stream_) {
    if (!x || !y) return SVPS_ERR_BAD_ARG;
    if (N <= 0 || HW <= 0 || C <= 0 || (C & 3) || C > 1024 || (256 % (C >> 2))) return SVPS_ERR_BAD_SHAPE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int tiles = (HW + 31) / 32;
    int wgs = (2048 + N - 1) / N;
    wgs = wgs < tiles ? wgs : tiles;
    const int tpw = (tiles + wgs - 1) / wgs;
    wgs = (tiles + tpw - 1) / tpw;
    const size_t lds = (size_t)32 * (C + 1) * sizeof(float);
    static SvpsLdsAttr attr;
    if (lds > 48 * 1024)
        if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(svps::nchw_to_pm_kernel), (int)lds); e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(svps::nchw_to_pm_kernel, dim3(wgs, N), dim3(256), lds, stream, x, y, HW, C, tpw);
    return (int)hipGetLastError();
}

extern "C" size_t svps_group_norm_relu_workspace_bytes(int N, int HW, int C) {
    if (N <= 0 || HW <= 0 || C <= 0) return 0;
    const int chunks = (HW + 2047) / 2048;
    return (size_t)N * chunks * 2 * C * sizeof(float) + (size_t)N * C * sizeof(float2);
}

extern "C" int svps_group_norm_relu_fwd(const float* x, const float* gamma, const float* beta, int groups, float eps, float* y,
                                        float* y_nchw, void* workspace, size_t workspace_bytes, int N, int HW, int C, void* stream_) {
    if (!y) return SVPS_ERR_BAD_ARG;
    return svps_group_norm_relu16_fwd(x, gamma, beta, groups, eps, y, y_nchw, nullptr, 0, workspace, workspace_bytes, N, HW, C, stream_);
}

extern "C" int svps_group_norm_relu16_fwd(const float* x, const float* gamma, const float* beta, int groups, float eps, float* y,
                                          float* y_nchw, void* y16, int y16_is_fp16, void* workspace, size_t workspace_bytes, int N,
                                          int HW, int C, void* stream_) {
    return svps_group_norm_relu_stats_fwd(x, nullptr, 0, gamma, beta, groups, eps, y, y_nchw, y16, y16_is_fp16, workspace, workspace_bytes,
                                          N, HW, C, stream_);
}

// partial != null: the per-channel sums come from the producer of x (svps_deform_conv_fused_stats_fwd: [N, chunks, 2, C]); the
// statistics pass over x is skipped and `workspace` only holds the per-channel scale / shift (N C float2)
extern "C" int svps_group_norm_relu_stats_fwd(const float* x, const float* partial_in, int chunks_in, const float* gamma, const float* beta,
                                              int groups, float eps, float* y, float* y_nchw, void* y16, int y16_is_fp16, void* workspace,
                                              size_t workspace_bytes, int N, int HW, int C, void* stream_) {
    if (!x || !gamma || !beta || !workspace || (!y && !y_nchw && !y16)) return SVPS_ERR_BAD_ARG;
    if (partial_in && chunks_in <= 0) return SVPS_ERR_BAD_SHAPE;
    if (N <= 0 || HW <= 0 || C <= 0 || (C & 3) || C > 1024 || groups <= 0 || groups > 256 || C % groups) return SVPS_ERR_BAD_SHAPE;
    if (256 % (C >> 2)) return SVPS_ERR_BAD_SHAPE;                           // float4 columns of a pixel row must divide the workgroup
    if (workspace_bytes < svps_group_norm_relu_workspace_bytes(N, HW, C)) return SVPS_ERR_WORKSPACE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int chunks = (HW + 2047) / 2048;
    float* own = static_cast<float*>(workspace);
    float2* ab = reinterpret_cast<float2*>(own + (size_t)N * chunks * 2 * C);
    const float* partial = own;
    if (!partial_in) hipLaunchKernelGGL(svps::gn_stats_kernel, dim3(chunks, N), dim3(256), 0, stream, x, own, HW, C, 2048);
    else hipLaunchKernelGGL(svps::gn_reduce_kernel, dim3(chunks, N), dim3(256), 0, stream, partial_in, own, C, chunks_in, chunks);
    hipLaunchKernelGGL(svps::gn_finalize_kernel, dim3(N), dim3(256), 0, stream, partial, gamma, beta, ab, HW, C, groups, chunks, eps);
    const int tiles = (HW + 31) / 32;
    int wgs = (2048 + N - 1) / N;                                           // ~2048 workgroups per launch
    wgs = wgs < tiles ? wgs : tiles;
    const int tpw = (tiles + wgs - 1) / wgs;
    wgs = (tiles + tpw - 1) / tpw;
    const size_t lds = y_nchw ? (size_t)32 * (C + 1) * sizeof(float) : 0;
    static SvpsLdsAttr attr;
    if (lds > 48 * 1024)
        if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(svps::gn_apply_kernel), (int)lds); e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(svps::gn_apply_kernel, dim3(wgs, N), dim3(256), lds, stream, x, (const float2*)ab, y, y_nchw, y16, y16_is_fp16, HW,
                       C, tpw);
    return (int)hipGetLastError();
}
