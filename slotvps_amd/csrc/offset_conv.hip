// The offset-producing convolution of DeformConvWithOffset (mmdet/ops/dcn/deform_conv.py: conv_offset, a plain 3 x 3 convolution,
// stride 1, padding 1, C -> 18 channels) on the pixel-major fp32 activations of the semantic tower.
// The framework ran it as an fp32 convolution on an NCHW copy of every layer's input (3.0 ms per T = 5 clip + the copies). Here:
//   out[n, o, y, x] = b[o] + sum_{tap, c} w[o, c, tap] x[n, y + ty - 1, x + tx - 1, c]         o < 32 (18 used), zero padding
// as a matrix-core product with the PIXELS as rows: a wave owns 32 consecutive pixels; the A fragment of a k-step (tap, 16 channels) is
// each lane's own 32 contiguous bytes of the shifted pixel's row (no LDS, nine-fold reuse through L1 / L2), split into bf16 hi + lo in
// registers; the B fragments are the packed weights [K / 16][2][64][8] (ops.pack_b_fragments of the [32, 9 C] matrix, k = tap C + c);
// three MFMAs per k-step (hi hi + lo hi + hi lo: fp32-class, like K7' and K8), fp32 accumulation, offsets written NCHW as K7' reads them.
// MEASURED SLOWER than the framework's convolution (3.6 against 3.0 ms per T = 5 clip at 1024 x 2048: a wave's A load touches 32 cache
// lines for 2 KiB; an LDS-tiled form would be needed), so UPSNetFPN.fuse_offset is OFF by default; kept as a tested option.
#include "common.h"
#include "../../include/slotvps_hip.h"

namespace svps {

constexpr int kOcPB = 1;                     // pixel blocks of 32 per wave (4: every weight fragment feeds four blocks - measured slower,
                                             // 4.6 against 3.6 ms per clip: the kernel is bound by its per-lane row loads, not by the weights)

struct OcFrag { f32x4 a[kOcPB][2]; u32x4 wh, wl; };

__global__ __launch_bounds__(256) void conv3x3_small_kernel(const float* __restrict__ x,        // [N, H, W, C]
                                                            const u32x4* __restrict__ wpack,    // [9 C / 16][2][64] x 16 B
                                                            const float* __restrict__ bias,     // [O] or null
                                                            float* __restrict__ out,            // [N, O, H, W]
                                                            int C, int H, int W, int O) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n = blockIdx.y;
    const int HW = H * W;
    const int p0 = (blockIdx.x * 4 + w) * (32 * kOcPB);
    if (p0 >= HW) return;
    int py[kOcPB], pxq[kOcPB];
#pragma unroll
    for (int b = 0; b < kOcPB; ++b) {
        const int p = p0 + 32 * b + r < HW ? p0 + 32 * b + r : HW - 1;
        py[b] = p / W;
        pxq[b] = p - py[b] * W;
    }
    const float* xn = x + (size_t)n * HW * C + 8 * h;
    const int kpt = C >> 4;                                   // k-steps per tap
    const int nks = 9 * kpt;
    f32x16 acc[kOcPB];
#pragma unroll
    for (int b = 0; b < kOcPB; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
    // fragment loads of k-step s (tap s / kpt, channels 16 (s % kpt) ..): issued one k-step ahead of their use
    auto load = [&](int s, OcFrag& f) {
        const int t = s / kpt, ks = s - t * kpt;
        const int ty = t / 3, tx = t - 3 * ty;
#pragma unroll
        for (int b = 0; b < kOcPB; ++b) {
            const int yy = py[b] + ty - 1, xx = pxq[b] + tx - 1;
            const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
            const float* row = xn + ((size_t)(ok ? yy : py[b]) * W + (ok ? xx : pxq[b])) * C + 16 * ks;
            f.a[b][0] = *reinterpret_cast<const f32x4*>(row);
            f.a[b][1] = *reinterpret_cast<const f32x4*>(row + 4);
            if (!ok) { f.a[b][0] = f32x4{0.f, 0.f, 0.f, 0.f}; f.a[b][1] = f.a[b][0]; }
        }
        f.wh = wpack[(size_t)s * 128 + lane];
        f.wl = wpack[(size_t)s * 128 + 64 + lane];
    };
    auto mma = [&](const OcFrag& f) {
        const bf16x8 wh = __builtin_bit_cast(bf16x8, f.wh), wl = __builtin_bit_cast(bf16x8, f.wl);
#pragma unroll
        for (int b = 0; b < kOcPB; ++b) {
            bf16x8 ah, al;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ah[j] = (__bf16)f.a[b][0][j];
                al[j] = (__bf16)(f.a[b][0][j] - (float)ah[j]);
                ah[4 + j] = (__bf16)f.a[b][1][j];
                al[4 + j] = (__bf16)(f.a[b][1][j] - (float)ah[4 + j]);
            }
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wh, acc[b], 0, 0, 0);
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wh, acc[b], 0, 0, 0);
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wl, acc[b], 0, 0, 0);
        }
    };
    OcFrag f0, f1;
    load(0, f0);
    for (int s = 0; s < nks; s += 2) {
        if (s + 1 < nks) load(s + 1, f1);
        mma(f0);
        if (s + 2 < nks) load(s + 2, f0);
        if (s + 1 < nks) mma(f1);
    }
    // register i = pixel row (i & 3) + 8 (i >> 2) + 4 h of a block's 32, column = lane r = output channel
    if (r < O) {
        const float bv = bias ? bias[r] : 0.f;
        float* on = out + ((size_t)n * O + r) * HW;
#pragma unroll
        for (int b = 0; b < kOcPB; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int q = p0 + 32 * b + 8 * g + 4 * h;
                if (q + 3 < HW && (HW & 3) == 0) {
                    *reinterpret_cast<f32x4*>(on + q) = f32x4{acc[b][4 * g] + bv, acc[b][4 * g + 1] + bv, acc[b][4 * g + 2] + bv, acc[b][4 * g + 3] + bv};
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (q + j < HW) on[q + j] = acc[b][4 * g + j] + bv;
                }
            }
    }
}

}  // namespace svps

extern "C" int svps_conv3x3_pm_small_fwd(const float* x_nhwc, const void* wpack, const float* bias, float* out, int N, int C, int H, int W,
                                         int O, void* stream_) {
    if (!x_nhwc || !wpack || !out) return SVPS_ERR_BAD_ARG;
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 15) || O <= 0 || O > 32) return SVPS_ERR_BAD_SHAPE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int waves = (H * W + 32 * svps::kOcPB - 1) / (32 * svps::kOcPB);
    hipLaunchKernelGGL(svps::conv3x3_small_kernel, dim3((waves + 3) / 4, N), dim3(256), 0, stream, x_nhwc, static_cast<const svps::u32x4*>(wpack),
                       bias, out, C, H, W, O);
    return (int)hipGetLastError();
}
