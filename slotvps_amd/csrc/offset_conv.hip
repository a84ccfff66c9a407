// The offset-producing convolution of DeformConvWithOffset (mmdet/ops/dcn/deform_conv.py: conv_offset, a plain 3 x 3 convolution,
// stride 1, padding 1, C -> 18 channels) on the pixel-major fp32 activations of the semantic tower.
// The framework ran it as an fp32 convolution on an NCHW copy of every layer's input (3.0 ms per T = 5 clip + the copies). Here:
//   out[n, o, y, x] = b[o] + sum_{tap, c} w[o, c, tap] x[n, y + ty - 1, x + tx - 1, c]         o < 32 (18 used), zero padding
// as a matrix-core product with the PIXELS as rows: a wave owns 32 consecutive pixels; the A fragment of a k-step (tap, 16 channels) is
// each lane's own 32 contiguous bytes of the shifted pixel's row (no LDS, nine-fold reuse through L1 / L2), split into bf16 hi + lo in
// registers; the B fragments are the packed weights [K / 16][2][64][8] (ops.pack_b_fragments of the [32, 9 C] matrix, k = tap C + c);
// three MFMAs per k-step (hi hi + lo hi + hi lo: fp32-class, like K7' and K8), fp32 accumulation, offsets written NCHW as K7' reads them.
// (That first form - v1 - measured slower than the framework's convolution; the shipped form is the LDS-tiled v3 below.)
#include "common.h"
#include "../../include/slotvps_hip.h"

namespace svps {

// v3 (the shipped form): LDS-tiled. A workgroup (4 waves) owns an 8 x 32 tile of output pixels; per chunk of 32 channels it stages the
// 10 x 34 halo of the tile ONCE (coalesced 16-byte loads, zero outside the image, split into bf16 hi / lo planes in LDS) and all nine taps
// read their shifted 32-pixel rows from it: the activations cross L2 -> CU 1.33 times instead of nine. Wave w computes output rows 2w, 2w+1
// of the tile: per chunk 9 taps x 2 k-steps x 2 rows x 3 MFMAs. (v1 / v2 - every lane loading its own shifted row segment from global
// memory per tap - measured 3.6 / 4.6 ms per clip against the framework's 3.0.)
constexpr int kOcTH = 8, kOcTW = 32, kOcCK = 32;
constexpr int kOcHR = kOcTH + 2, kOcHC = kOcTW + 2;
constexpr int kOcPix = kOcCK * 2 + 16;                         // bytes per halo pixel and plane (64 B of channels + pad)
constexpr int kOcPlane = kOcHR * kOcHC * kOcPix;
constexpr int kOcItems = kOcHR * kOcHC * (kOcCK / 4);          // 16-byte fp32 segments of a halo chunk
constexpr int kOcPerThread = (kOcItems + 255) / 256;

__global__ __launch_bounds__(256, 2) void conv3x3_small_kernel(const float* __restrict__ x,        // [N, H, W, C]
                                                               const u32x4* __restrict__ wpack,    // [9 C / 16][2][64] x 16 B
                                                               const float* __restrict__ bias,     // [O] or null
                                                               float* __restrict__ out,            // [N, O, H, W]
                                                               int C, int H, int W, int O) {
    __shared__ __attribute__((aligned(16))) char smem[2 * kOcPlane];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n = blockIdx.z;
    const int ty0 = blockIdx.y * kOcTH, tx0 = blockIdx.x * kOcTW;
    const float* xn = x + (size_t)n * H * W * C;
    const int kpt = C >> 4;
    f32x16 acc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
    for (int c0 = 0; c0 < C; c0 += kOcCK) {
        // ---- halo chunk -> LDS
        f32x4 v[kOcPerThread];
#pragma unroll
        for (int k = 0; k < kOcPerThread; ++k) {
            const int it = tid + 256 * k;
            const int px = it >> 3, seg = it & 7;
            const int hy = px / kOcHC, hx = px - hy * kOcHC;
            const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
            const bool ok = it < kOcItems && gy >= 0 && gy < H && gx >= 0 && gx < W;
            v[k] = ok ? *reinterpret_cast<const f32x4*>(xn + ((size_t)gy * W + gx) * C + c0 + 4 * seg) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();                                           // the previous chunk's fragment reads are done
#pragma unroll
        for (int k = 0; k < kOcPerThread; ++k) {
            const int it = tid + 256 * k;
            if (it < kOcItems) {
                const int px = it >> 3, seg = it & 7;
                bf16x4 vh, vl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    vh[e] = (__bf16)v[k][e];
                    vl[e] = (__bf16)(v[k][e] - (float)vh[e]);
                }
                *reinterpret_cast<bf16x4*>(smem + px * kOcPix + seg * 8) = vh;
                *reinterpret_cast<bf16x4*>(smem + kOcPlane + px * kOcPix + seg * 8) = vl;
            }
        }
        __syncthreads();
        // ---- nine taps x two k-steps on the staged halo
        const int s0 = c0 >> 4;
        for (int t = 0; t < 9; ++t) {
            const int ty = t / 3, tx = t - 3 * ty;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const u32x4* wf = wpack + (size_t)(t * kpt + s0 + ks) * 128 + lane;
                const bf16x8 wh = __builtin_bit_cast(bf16x8, wf[0]);
                const bf16x8 wl = __builtin_bit_cast(bf16x8, wf[64]);
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int off = ((2 * w + b + ty) * kOcHC + r + tx) * kOcPix + (16 * ks + 8 * h) * 2;
                    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(smem + off);
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(smem + kOcPlane + off);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wh, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wh, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wl, acc[b], 0, 0, 0);
                }
            }
        }
    }
    // register i of block b = pixel x = tx0 + (i & 3) + 8 (i >> 2) + 4 h of output row ty0 + 2 w + b, column = lane r = output channel
    if (r < O) {
        const float bv = bias ? bias[r] : 0.f;
        const size_t HW = (size_t)H * W;
        float* on = out + ((size_t)n * O + r) * HW;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int gy = ty0 + 2 * w + b;
            if (gy >= H) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int gx = tx0 + 8 * g + 4 * h;
                float* dst = on + (size_t)gy * W + gx;
                if (gx + 3 < W && (W & 3) == 0) {
                    *reinterpret_cast<f32x4*>(dst) = f32x4{acc[b][4 * g] + bv, acc[b][4 * g + 1] + bv, acc[b][4 * g + 2] + bv, acc[b][4 * g + 3] + bv};
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (gx + j < W) dst[j] = acc[b][4 * g + j] + bv;
                }
            }
        }
    }
}

}  // namespace svps

extern "C" int svps_conv3x3_pm_small_fwd(const float* x_nhwc, const void* wpack, const float* bias, float* out, int N, int C, int H, int W,
                                         int O, void* stream_) {
    if (!x_nhwc || !wpack || !out) return SVPS_ERR_BAD_ARG;
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % svps::kOcCK) || O <= 0 || O > 32) return SVPS_ERR_BAD_SHAPE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(svps::conv3x3_small_kernel, dim3((W + svps::kOcTW - 1) / svps::kOcTW, (H + svps::kOcTH - 1) / svps::kOcTH, N), dim3(256), 0,
                       stream, x_nhwc, static_cast<const svps::u32x4*>(wpack), bias, out, C, H, W, O);
    return (int)hipGetLastError();
}
