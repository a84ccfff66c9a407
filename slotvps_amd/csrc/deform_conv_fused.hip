// K7' - deformable convolution forward (DCNv1) WITHOUT a column buffer, for gfx950.
//
// Replaces deform_conv_forward_cuda of the reference (mmdet/ops/dcn/src/deform_conv_cuda.cpp:152-258): the reference samples
// the 3 x 3 bilinear taps into a [C * 9, N * Ho * Wo] column buffer (deformable_im2col_gpu_kernel,
// deform_conv_cuda_kernel.cu:190-241, bilinear rule :82-114) and multiplies it with the flattened weight (addmm_).
// Here the sampled taps never leave the CU: a workgroup owns 128 output pixels, gathers one (tap, 64-channel) chunk of
// their samples at a time into LDS as a matrix-core B tile and accumulates  out[o, px] += W[o, tap, c] * sample[px, tap, c]
// on v_mfma_f32_32x32x16_bf16.
//
// Precision: the reference's op is fp32. Both operands are carried as bf16 hi + lo (16-bit mantissa) and the three
// significant products (hi.hi, lo.hi, hi.lo) are accumulated in fp32: "split-bf16", relative error ~1e-5 of the largest
// term - the class of the fp32 op's own summation-order noise - at a third of the bf16 matrix rate instead of the fp32
// matrix rate, which on gfx950 is the vector rate (1/16).
//
// Layouts: input pixel-major NHWC fp32 (a pixel's 64-channel chunk is 256 contiguous bytes: coalesced corner reads);
// offsets in the reference's layout [N, 2*kh*kw, Ho, Wo] (channel 2t = dy, 2t+1 = dx of tap t); weights pre-packed by the
// host into MFMA A-fragment order (hi and lo), k = tap * C + c; output pixel-major [N, Ho*Wo, O] fp32.
// Sampling rule (:82-114, :222-236): a sample outside (-1, H) x (-1, W) is zero; corners outside the image contribute zero.
//
// Mapping: 8 waves. O = 256: wave w owns output channels [32w, 32w+32) for all four 32-pixel blocks of the tile;
// O = 128: wave w owns channel block w & 3 for pixel blocks 2(w>>2), 2(w>>2)+1. Per chunk and wave 48 (24) MFMA, then the
// gather of chunk i+1 (16 x 16-byte loads per thread), its blend + hi/lo split + LDS write; one workgroup barrier per chunk;
// weights stream from L2 in fragment order (1 KiB per wave instruction).
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "../../include/slotvps_hip.h"

namespace svps {

// Two tile shapes (round 4), chosen per launch (svps_deform_conv_fused_stats_fwd): 128 output pixels per workgroup in chunks of 64 channels
// (4 k-steps), or 256 pixels in chunks of 32 channels (2 k-steps) - the same LDS budget and the same MFMAs / gathers per chunk, but every
// weight fragment streamed from L2 feeds twice the pixels at twice the barriers per MAC. Measured (tools/kbench_k7.py, T = 5):
// O = 128 at 256 x 512: 2 425 against 2 630 us; O = 256 at 256 x 512: 3 486 against 3 350; 64 x 128 (320 against 160 workgroups on 256
// CUs): 244 against 352; 32 x 64 (80 against 40 workgroups): 134 against 233.
template <bool T256>
struct DcTile {
    static constexpr int px = T256 ? 256 : 128;        // output pixels per workgroup
    static constexpr int ch = T256 ? 32 : 64;          // channels per chunk
    static constexpr int ks = ch / 16;                 // k-steps per chunk
    static constexpr int gr = ch / 4;                  // 4-channel groups per pixel and chunk
    static constexpr int row = ch * 2 + 16;            // bytes per pixel row of a sample tile (padded: conflict-free 16-byte fragment reads)
};

template <bool T256>
struct DcLds {
    static constexpr int coords = 0;                                   // [9][px] x {int idx[4]; float w[4]}
    static constexpr int bufs = 9 * DcTile<T256>::px * 32;             // [2][hi | lo][px][row]
    static constexpr int buf_bytes = 2 * DcTile<T256>::px * DcTile<T256>::row;
    static constexpr int total = bufs + 2 * buf_bytes;
};
static_assert(DcLds<true>::total <= 160 * 1024 && DcLds<false>::total <= 160 * 1024, "LDS budget");

template <int OB, bool T256>               // output channel blocks: 8 (O = 256) or 4 (O = 128); tile shape
__global__ __launch_bounds__(512) void deform_conv_fused_kernel(const float* __restrict__ x,        // [N, H, W, C]
                                                                const float* __restrict__ offset,   // [N, 18, Ho, Wo]
                                                                const __bf16* __restrict__ wpack,   // [OB, KS, 2, 64, 8]
                                                                float* __restrict__ out,            // [N, Ho*Wo, 32 OB]
                                                                int C, int H, int W, int Ho, int Wo, int pad, int stride, int dil,
                                                                float* __restrict__ gn_part = nullptr) {   // [N, chunks, 2, 32 OB] or null
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using Lds = DcLds<T256>;
    constexpr int kDcPx = DcTile<T256>::px, kDcCh = DcTile<T256>::ch, kDcKs = DcTile<T256>::ks, kDcGr = DcTile<T256>::gr, kDcRow = DcTile<T256>::row;
    constexpr int NB = 32 / OB;            // pixel blocks per wave: 4 or ... (OB = 8 -> 4, OB = 4 -> 2)
#ifndef SVPS_K7_O128_FOUR_WAVES
#define SVPS_K7_O128_FOUR_WAVES 1
#endif
    // O = 128, round 4: waves 0 .. 3 own one output block each for ALL four pixel blocks and waves 4 .. 7 only gather / blend (before:
    // wave w and w + 4 shared an output block for two pixel blocks each and both streamed its weight fragments from L2 - twice the
    // weight traffic of the O = 256 form per output; tools/kbench_k7.py)
    constexpr bool kFourMma = OB == 4 && SVPS_K7_O128_FOUR_WAVES;
    constexpr int NPB = (OB == 8 || kFourMma) ? kDcPx / 32 : kDcPx / 64;
    static_assert(OB == 8 || OB == 4, "O = 256 or 128");
    (void)NB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int n = blockIdx.y;
    const int HWo = Ho * Wo;
    const int p0 = blockIdx.x * kDcPx;
    const int ob = (OB == 8) ? w : (w & 3);
    const int pb0 = (OB == 8 || kFourMma) ? 0 : (kDcPx / 64) * (w >> 2);
    const bool mma_wave = !kFourMma || w < 4;
    const int KS = 9 * C / 16;
    const int cchunks = C / kDcCh, nch = 9 * cchunks;
    const float* xn = x + (size_t)n * H * W * C;

    // ---- sampling coordinates of the tile: 9 taps x 128 pixels -> 4 clamped corner indices + 4 weights (0 where invalid) ----
    for (int e = tid; e < 9 * kDcPx; e += 512) {
        const int t = e / kDcPx, px = e - t * kDcPx;
        int p = p0 + px;
        p = p < HWo ? p : HWo - 1;
        const int ho = p / Wo, wo = p - ho * Wo;
        const float* off = offset + ((size_t)n * 18 + 2 * t) * HWo + p;
        const int i = t / 3, j = t - 3 * i;
        const float hf = (float)(ho * stride - pad + i * dil) + off[0];
        const float wf = (float)(wo * stride - pad + j * dil) + off[HWo];
        int idx[4] = {0, 0, 0, 0};
        float wt[4] = {0.f, 0.f, 0.f, 0.f};
        if (hf > -1.f && wf > -1.f && hf < (float)H && wf < (float)W) {
            const int hl = (int)floorf(hf), wl = (int)floorf(wf);
            const int hh = hl + 1, wh = wl + 1;
            const float lh = hf - (float)hl, lw = wf - (float)wl;
            const float uh = 1.f - lh, uw = 1.f - lw;
            const int hlc = hl < 0 ? 0 : hl, wlc = wl < 0 ? 0 : wl;
            const int hhc = hh > H - 1 ? H - 1 : hh, whc = wh > W - 1 ? W - 1 : wh;
            idx[0] = hlc * W + wlc; idx[1] = hlc * W + whc; idx[2] = hhc * W + wlc; idx[3] = hhc * W + whc;
            wt[0] = (hl >= 0 && wl >= 0) ? uh * uw : 0.f;
            wt[1] = (hl >= 0 && wh <= W - 1) ? uh * lw : 0.f;
            wt[2] = (hh <= H - 1 && wl >= 0) ? lh * uw : 0.f;
            wt[3] = (hh <= H - 1 && wh <= W - 1) ? lh * lw : 0.f;
        }
        int* ci = reinterpret_cast<int*>(smem + Lds::coords + e * 32);
        ci[0] = idx[0]; ci[1] = idx[1]; ci[2] = idx[2]; ci[3] = idx[3];
        float* cw = reinterpret_cast<float*>(ci + 4);
        cw[0] = wt[0]; cw[1] = wt[1]; cw[2] = wt[2]; cw[3] = wt[3];
    }
    __syncthreads();

    // ---- gather / blend: thread -> items q = tid + 512 i (i < 4): pixel q >> 4, channel group (q & 15) * 4 of the chunk ----
    f32x4 gv[4][4];                       // [item][corner]
    f32x4 gw[4];                          // corner weights of the item's (tap, pixel)
    auto gather = [&](int ch) {
        const int t = ch / cchunks, c0 = (ch - t * cchunks) * kDcCh;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + 512 * i, px = q / kDcGr, cg = (q % kDcGr) * 4;
            const int* ci = reinterpret_cast<const int*>(smem + Lds::coords + (t * kDcPx + px) * 32);
            const u32x4 id = *reinterpret_cast<const u32x4*>(ci);
            gw[i] = *reinterpret_cast<const f32x4*>(ci + 4);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                gv[i][k] = *reinterpret_cast<const f32x4*>(xn + (size_t)id[k] * C + c0 + cg);
        }
    };
    auto blend_store = [&](int buf) {
        char* bh = smem + Lds::bufs + buf * Lds::buf_bytes;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + 512 * i, px = q / kDcGr, cg = (q % kDcGr) * 4;
            bf16x4 vh, vl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // the reference's expression: w1 v1 + w2 v2 + w3 v3 + w4 v4 (:112)
                const float v = gw[i][0] * gv[i][0][e] + gw[i][1] * gv[i][1][e] + gw[i][2] * gv[i][2][e] + gw[i][3] * gv[i][3][e];
                vh[e] = (__bf16)v;
                vl[e] = (__bf16)(v - (float)vh[e]);
            }
            *reinterpret_cast<bf16x4*>(bh + px * kDcRow + cg * 2) = vh;
            *reinterpret_cast<bf16x4*>(bh + kDcPx * kDcRow + px * kDcRow + cg * 2) = vl;
        }
    };

    f32x16 acc[NPB];
#pragma unroll
    for (int b = 0; b < NPB; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;

    gather(0);
    blend_store(0);
    __syncthreads();
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(wpack) + ((size_t)ob * KS * 2) * 64 + lane;   // fragment (ks, part): + (ks * 2 + part) * 64
    for (int ch = 0; ch < nch; ++ch) {
        const char* bh = smem + Lds::bufs + (ch & 1) * Lds::buf_bytes;
        const char* bl = bh + kDcPx * kDcRow;
        // weights of the chunk's four k-steps (hi, lo): k-step index of chunk ch = ch * 4 + u  (k = tap * C + c)
        bf16x8 ah[kDcKs], al[kDcKs];
#ifndef SVPS_K7_ABL
#define SVPS_K7_ABL 0          // timing-only ablations (wrong results; separate library): 1 weights of chunk 0 every time (L2-resident), 2 no gather
#endif
        const int chw = (SVPS_K7_ABL & 1) ? 0 : ch;
        if (mma_wave) {
#pragma unroll
        for (int u = 0; u < kDcKs; ++u) {
            ah[u] = __builtin_bit_cast(bf16x8, wsrc[(size_t)((chw * kDcKs + u) * 2) * 64]);
            al[u] = __builtin_bit_cast(bf16x8, wsrc[(size_t)((chw * kDcKs + u) * 2 + 1) * 64]);
        }
#pragma unroll
        for (int b = 0; b < NPB; ++b) {
            const int prow = (32 * (pb0 + b) + r) * kDcRow + 16 * h;
            bf16x8 sh[kDcKs], sl[kDcKs];
#pragma unroll
            for (int u = 0; u < kDcKs; ++u) {
                sh[u] = *reinterpret_cast<const bf16x8*>(bh + prow + 32 * u);
                sl[u] = *reinterpret_cast<const bf16x8*>(bl + prow + 32 * u);
            }
#pragma unroll
            for (int u = 0; u < kDcKs; ++u) {
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[u], sh[u], acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[u], sh[u], acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[u], sl[u], acc[b], 0, 0, 0);
            }
        }
        }
        // The gather of the next chunk is issued AFTER this chunk's MFMAs (they run on while it waits for memory; the partner
        // wave of the SIMD fills the gaps). Issued before them - 80 registers of loaded samples live across the MFMA
        // section - the hipcc build of this kernel returned different, wrong tiles from run to run on gfx950 (every load was
        // complete before the first MFMA by the compiler's own vmcnt waits, an explicit vmcnt(0) changed nothing, no
        // spills); with this order the result is bitwise reproducible and equal to the column-buffer path to 8e-6, at the
        // same speed. tests/test_deform_conv.py asserts both.
        if (ch + 1 < nch) {
            if (!(SVPS_K7_ABL & 2)) gather(ch + 1);
            blend_store((ch + 1) & 1);
        }
        __syncthreads();
    }

    // ---- per-channel sums and sums of squares of this workgroup's pixels for the GroupNorm behind the layer (round 4: its statistics
    // pass re-read the whole result): every wave reduces its pixel blocks over the 32 lanes that hold the same channels and writes ONE
    // partial row per (tile, pixel half) - chunks = tiles (O = 256) or 2 tiles (O = 128) per frame, each (chunk, channel) written by
    // exactly one lane: deterministic, combined in float64 by gn_finalize_kernel
    if (gn_part && mma_wave) {
        float s1[16], s2[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
#pragma unroll
        for (int b = 0; b < NPB; ++b) {
            if (p0 + 32 * (pb0 + b) + r < HWo) {
#pragma unroll
                for (int i = 0; i < 16; ++i) { s1[i] += acc[b][i]; s2[i] += acc[b][i] * acc[b][i]; }
            }
        }
#pragma unroll
        for (int m = 16; m > 0; m >>= 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { s1[i] += __shfl_xor(s1[i], m); s2[i] += __shfl_xor(s2[i], m); }
        }
        if (r == 0) {
            const int chunks = (OB == 8 || kFourMma) ? gridDim.x : 2 * gridDim.x;
            const int chunk = (OB == 8 || kFourMma) ? blockIdx.x : 2 * blockIdx.x + (w >> 2);
            float* base = gn_part + (((size_t)n * chunks + chunk) * 2) * (32 * OB) + 32 * ob + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                *reinterpret_cast<f32x4*>(base + 8 * g) = f32x4{s1[4 * g], s1[4 * g + 1], s1[4 * g + 2], s1[4 * g + 3]};
                *reinterpret_cast<f32x4*>(base + 32 * OB + 8 * g) = f32x4{s2[4 * g], s2[4 * g + 1], s2[4 * g + 2], s2[4 * g + 3]};
            }
        }
    }
    // ---- out[n, p, o]: lane (pixel r of block b, h) holds output channels 32 ob + 8 g + 4 h + i ----
    float* on = out + (size_t)n * HWo * (32 * OB);
    if (!mma_wave) return;
#pragma unroll
    for (int b = 0; b < NPB; ++b) {
        const int p = p0 + 32 * (pb0 + b) + r;
        if (p < HWo) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {acc[b][4 * g], acc[b][4 * g + 1], acc[b][4 * g + 2], acc[b][4 * g + 3]};
                *reinterpret_cast<f32x4*>(on + (size_t)p * (32 * OB) + 32 * ob + 8 * g + 4 * h) = v;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// K7'' (round 5) - the same function with the input staged through LDS: the form above gathers 4 corners x 9 taps = 36 rows of 4 C
// bytes per output pixel from L2 (24 GB for the 256 -> 128 layer of the semantic tower at 256 x 512, T = 5: L2-bandwidth-bound,
// 2.4 ms). Here a workgroup owns an 8 x 16 block of output pixels; per 32-channel chunk the (8 + 7) x (16 + 7) input pixels its samples
// can touch when |offset| <= 2 (one tap to each side, two pixels of offset, the bilinear neighbour) are loaded ONCE into LDS
// (requested under the previous chunk's last taps into six registers) and the nine taps of the chunk gather their corners from there:
// 1.6 instead of 36 pixel rows per output pixel and chunk from L2. A SAMPLE with a corner outside the staged region (a larger offset)
// is gathered from global memory as before - decided per sample in the coordinate pass, so the result never depends on it. Loop order:
// channel chunk outer, tap inner (the accumulation order differs from the form above by fp32 rounding only); 3 x 3, stride 1, pad 1,
// dilation 1 (the layers of the semantic tower), C % 32 == 0.
struct DhLds {
    static constexpr int TH = 8, TW = 16, PX = 128, CH = 32, KS = 2, GR = 8;
    static constexpr int ROW = CH * 2 + 16;                       // bytes per pixel row of a sample tile (padded)
    static constexpr int D = 2;                                   // |offset| covered by the staged region
    static constexpr int RH = TH + 2 + 2 * D + 1, RW = TW + 2 + 2 * D + 1;   // 15 x 23 input pixels
    static constexpr int RS = CH * 4 + 16;                        // bytes per staged pixel (fp32, padded)
    static constexpr int RITEMS = RH * RW * 8;                    // 16-byte pieces of a staged region
    static constexpr int RPT = (RITEMS + 511) / 512;              // ... per thread: 6
    static constexpr int coords = 0;                              // [9][PX] x {int idx[4]; float w[4]}
    static constexpr int bufs = 9 * PX * 32;                      // [2][hi | lo][PX][ROW]
    static constexpr int buf_bytes = 2 * PX * ROW;
    static constexpr int region = bufs + 2 * buf_bytes;
    static constexpr int total = region + RH * RW * RS;
};
static_assert(DhLds::total <= 160 * 1024, "LDS budget");

template <int OB>                          // output channel blocks: 8 (O = 256) or 4 (O = 128: waves 0 .. 3 multiply, 4 .. 7 only gather)
__global__ __launch_bounds__(512) void deform_conv_halo_kernel(const float* __restrict__ x,        // [N, H, W, C]
                                                               const float* __restrict__ offset,   // [N, 18, H, W]
                                                               const __bf16* __restrict__ wpack,   // [OB, KS, 2, 64, 8]
                                                               float* __restrict__ out,            // [N, H*W, 32 OB]
                                                               int C, int H, int W, int tiles_x,
                                                               float* __restrict__ gn_part) {      // [N, tiles, 2, 32 OB] or null
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using Lds = DhLds;
    constexpr int PX = Lds::PX, CH = Lds::CH, KS = Lds::KS, GR = Lds::GR, ROW = Lds::ROW, RW = Lds::RW, RH = Lds::RH, RS = Lds::RS;
    constexpr int NPB = PX / 32;
    static_assert(OB == 8 || OB == 4, "O = 256 or 128");
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int n = blockIdx.y;
    const int HW = H * W;
    const int tyb = blockIdx.x / tiles_x, txb = blockIdx.x - tyb * tiles_x;
    const int y0 = tyb * Lds::TH, x0 = txb * Lds::TW;
    const int ry0 = y0 - 1 - Lds::D, rx0 = x0 - 1 - Lds::D;
    const int ob = (OB == 8) ? w : (w & 3);
    const bool mma_wave = OB == 8 || w < 4;
    const int KSall = 9 * C / 16;
    const int cchunks = C / CH, nch = 9 * cchunks;
    const float* xn = x + (size_t)n * HW * C;

    // ---- sampling coordinates of the tile: 9 taps x 128 pixels -> 4 corner addresses + 4 weights (0 where invalid). A sample whose four
    // corners lie inside the staged region gets their byte offsets into it (>= 0); any other sample (|offset| > 2) the pixel indices of the
    // frame encoded as -(index + 1) and is gathered from global memory - per SAMPLE, so one far sample costs one far gather, not the tile
    for (int e = tid; e < 9 * PX; e += 512) {
        const int t = e / PX, px = e - t * PX;
        int ho = y0 + (px >> 4), wo = x0 + (px & 15);
        ho = ho < H ? ho : H - 1;
        wo = wo < W ? wo : W - 1;
        const float* off = offset + ((size_t)n * 18 + 2 * t) * HW + (size_t)ho * W + wo;
        const int i = t / 3, j = t - 3 * i;
        const float hf = (float)(ho - 1 + i) + off[0];
        const float wf = (float)(wo - 1 + j) + off[HW];
        int idx[4] = {0, 0, 0, 0};
        float wt[4] = {0.f, 0.f, 0.f, 0.f};
        if (hf > -1.f && wf > -1.f && hf < (float)H && wf < (float)W) {
            const int hl = (int)floorf(hf), wl = (int)floorf(wf);
            const int hh = hl + 1, wh = wl + 1;
            const float lh = hf - (float)hl, lw = wf - (float)wl;
            const float uh = 1.f - lh, uw = 1.f - lw;
            const int hlc = hl < 0 ? 0 : hl, wlc = wl < 0 ? 0 : wl;
            const int hhc = hh > H - 1 ? H - 1 : hh, whc = wh > W - 1 ? W - 1 : wh;
            wt[0] = (hl >= 0 && wl >= 0) ? uh * uw : 0.f;
            wt[1] = (hl >= 0 && wh <= W - 1) ? uh * lw : 0.f;
            wt[2] = (hh <= H - 1 && wl >= 0) ? lh * uw : 0.f;
            wt[3] = (hh <= H - 1 && wh <= W - 1) ? lh * lw : 0.f;
            const int a = hlc - ry0, b = hhc - ry0, c = wlc - rx0, d = whc - rx0;
            if (a >= 0 && b < RH && c >= 0 && d < RW && a < RH && b >= 0 && c < RW && d >= 0) {
                idx[0] = (a * RW + c) * RS; idx[1] = (a * RW + d) * RS; idx[2] = (b * RW + c) * RS; idx[3] = (b * RW + d) * RS;
            } else {
                idx[0] = -(hlc * W + wlc) - 1; idx[1] = -(hlc * W + whc) - 1; idx[2] = -(hhc * W + wlc) - 1; idx[3] = -(hhc * W + whc) - 1;
            }
        }
        int* ci = reinterpret_cast<int*>(smem + Lds::coords + e * 32);
        ci[0] = idx[0]; ci[1] = idx[1]; ci[2] = idx[2]; ci[3] = idx[3];
        float* cw = reinterpret_cast<float*>(ci + 4);
        cw[0] = wt[0]; cw[1] = wt[1]; cw[2] = wt[2]; cw[3] = wt[3];
    }
    __syncthreads();

    // ---- staged region of one 32-channel chunk: RH x RW pixels x 128 B, thread -> 16-byte pieces tid + 512 i ----
    f32x4 rg[Lds::RPT];
    auto region_fetch = [&](int cc) {
#pragma unroll
        for (int i = 0; i < Lds::RPT; ++i) {
            const int j = tid + 512 * i, rp = j >> 3, l8 = j & 7;
            const int ry = rp / RW, rx = rp - ry * RW;
            const int y = ry0 + ry, xx = rx0 + rx;
            rg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (j < Lds::RITEMS && y >= 0 && y < H && xx >= 0 && xx < W)
                rg[i] = *reinterpret_cast<const f32x4*>(xn + ((size_t)y * W + xx) * C + cc * CH + 4 * l8);
        }
    };
    auto region_commit = [&]() {
#pragma unroll
        for (int i = 0; i < Lds::RPT; ++i) {
            const int j = tid + 512 * i, rp = j >> 3, l8 = j & 7;
            if (j < Lds::RITEMS) *reinterpret_cast<f32x4*>(smem + Lds::region + rp * RS + 16 * l8) = rg[i];
        }
    };

    // ---- gather / blend: 1024 items per chunk (pixel q >> 3, channel group (q & 7) * 4). O = 256: every thread two (q = tid + 512 i).
    // O = 128: the four multiplying waves carry the chunk's 24 MFMAs and the requests of the next chunk's weight fragments, so their threads
    // take ONE item (q = tid) and the threads of waves 4 .. 7 THREE (q = tid + 256 i): measured at 256 x 512, 256 -> 128: equal shares
    // 1 917 us, one item on the multiplying waves 1 790, none the same within the box-to-box noise. The gather / blend is VALU-bound
    // (~150 vector instructions per item): packed fp32 math below
#ifndef SVPS_K7_MMA_ITEMS
#define SVPS_K7_MMA_ITEMS 1            // O = 128: items per thread of the multiplying waves (0 or 1; the other waves take the rest)
#endif
    constexpr int NI = OB == 8 ? 2 : 4 - SVPS_K7_MMA_ITEMS;
    const int ni = OB == 8 ? 2 : (mma_wave ? SVPS_K7_MMA_ITEMS : 4 - SVPS_K7_MMA_ITEMS);
    auto item = [&](int i) { return OB == 8 ? tid + 512 * i : (mma_wave ? tid : (SVPS_K7_MMA_ITEMS ? tid : tid - 256) + 256 * i); };
    f32x4 gv[NI][4];                      // [item][corner]
    f32x4 gw[NI];                         // corner weights of the item's (tap, pixel)
    auto gather = [&](int t, int cc) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (i < ni) {
                const int q = item(i), px = q / GR, cg = (q % GR) * 4;
                const int* ci = reinterpret_cast<const int*>(smem + Lds::coords + (t * PX + px) * 32);
                const u32x4 id = *reinterpret_cast<const u32x4*>(ci);
                gw[i] = *reinterpret_cast<const f32x4*>(ci + 4);
                if ((int)id[0] < 0) {                      // a far sample: its corners from global memory
#pragma unroll
                    for (int k = 0; k < 4; ++k) gv[i][k] = *reinterpret_cast<const f32x4*>(xn + (size_t)(-(int)id[k] - 1) * C + cc * CH + cg);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) gv[i][k] = *reinterpret_cast<const f32x4*>(smem + Lds::region + id[k] + cg * 4);
                }
            }
        }
    };
    auto blend_store = [&](int buf) {
        char* bh = smem + Lds::bufs + buf * Lds::buf_bytes;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (i < ni) {
                const int q = item(i), px = q / GR, cg = (q % GR) * 4;
                // the reference's expression: w1 v1 + w2 v2 + w3 v3 + w4 v4 (:112), two channels per packed instruction (v_pk_mul / v_pk_fma_f32;
                // these waves' vector work is what bounds the kernel, and nothing of it sits behind an MFMA of its own wave)
                typedef float f32x2_t __attribute__((ext_vector_type(2)));
                typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                bf16x2_t vh[2], vl[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const f32x2_t c0 = {gv[i][0][2 * e], gv[i][0][2 * e + 1]}, c1 = {gv[i][1][2 * e], gv[i][1][2 * e + 1]};
                    const f32x2_t c2 = {gv[i][2][2 * e], gv[i][2][2 * e + 1]}, c3 = {gv[i][3][2 * e], gv[i][3][2 * e + 1]};
                    const f32x2_t w0 = {gw[i][0], gw[i][0]}, w1 = {gw[i][1], gw[i][1]}, w2 = {gw[i][2], gw[i][2]}, w3 = {gw[i][3], gw[i][3]};
                    const f32x2_t v = __builtin_elementwise_fma(w3, c3, __builtin_elementwise_fma(w2, c2, __builtin_elementwise_fma(w1, c1, w0 * c0)));
                    vh[e] = __builtin_convertvector(v, bf16x2_t);
                    vl[e] = __builtin_convertvector(v - __builtin_convertvector(vh[e], f32x2_t), bf16x2_t);
                }
                *reinterpret_cast<bf16x4*>(bh + px * ROW + cg * 2) = __builtin_shufflevector(vh[0], vh[1], 0, 1, 2, 3);
                *reinterpret_cast<bf16x4*>(bh + PX * ROW + px * ROW + cg * 2) = __builtin_shufflevector(vl[0], vl[1], 0, 1, 2, 3);
            }
        }
    };

    f32x16 acc[NPB];
#pragma unroll
    for (int b = 0; b < NPB; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;

    region_fetch(0);
    region_commit();
    __syncthreads();
    gather(0, 0);
    blend_store(0);
    __syncthreads();
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(wpack) + ((size_t)ob * KSall * 2) * 64 + lane;   // fragment (ks, part): + (ks * 2 + part) * 64
    int t = 0, cc = 0;                                         // chunk ch = (channel chunk cc, tap t): cc outer, t inner
    // weight fragments of a chunk: k-steps of (tap t, channels 32 cc ..), k = tap * C + c; requested ONE CHUNK AHEAD (they come from L2: used
    // in the chunk they were requested in, their latency sat at the head of every chunk's MFMA section)
    bf16x8 ahn[KS], aln[KS];
    auto weights = [&](int t_, int cc_) {
        const int ks0 = (t_ * C + cc_ * CH) / 16;
#pragma unroll
        for (int u = 0; u < KS; ++u) {
            ahn[u] = __builtin_bit_cast(bf16x8, wsrc[(size_t)((ks0 + u) * 2) * 64]);
            aln[u] = __builtin_bit_cast(bf16x8, wsrc[(size_t)((ks0 + u) * 2 + 1) * 64]);
        }
    };
    if (mma_wave) weights(0, 0);
    for (int ch = 0; ch < nch; ++ch) {
        const char* bh = smem + Lds::bufs + (ch & 1) * Lds::buf_bytes;
        const char* bl = bh + PX * ROW;
        if (mma_wave) {
            bf16x8 ah[KS], al[KS];
#pragma unroll
            for (int u = 0; u < KS; ++u) { ah[u] = ahn[u]; al[u] = aln[u]; }
            if (ch + 1 < nch) weights(t == 8 ? 0 : t + 1, t == 8 ? cc + 1 : cc);
#pragma unroll
            for (int b = 0; b < NPB; ++b) {
                const int prow = (32 * b + r) * ROW + 16 * h;
                bf16x8 sh[KS], sl[KS];
#pragma unroll
                for (int u = 0; u < KS; ++u) {
                    sh[u] = *reinterpret_cast<const bf16x8*>(bh + prow + 32 * u);
                    sl[u] = *reinterpret_cast<const bf16x8*>(bl + prow + 32 * u);
                }
#pragma unroll
                for (int u = 0; u < KS; ++u) {
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[u], sh[u], acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[u], sh[u], acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[u], sl[u], acc[b], 0, 0, 0);
                }
            }
        }
        // next chunk (issued AFTER this chunk's MFMAs, as in the form above)
        int tn = t + 1, ccn = cc;
        if (tn == 9) { tn = 0; ++ccn; }
        if (ch + 1 < nch) {
            if (t == 4 && cc + 1 < cchunks) region_fetch(cc + 1);            // lands under the taps 5 .. 8 of this channel chunk
            if (t == 8) {
                // the gathers of this channel chunk are all done (the last one ran in the previous iteration, before its barrier)
                region_commit();
                __syncthreads();
            }
            gather(tn, ccn);
            blend_store((ch + 1) & 1);
        }
        t = tn;
        cc = ccn;
        __syncthreads();
    }

    // ---- GroupNorm partial sums (see the form above): one partial row per tile and multiplying wave ----
    if (!mma_wave) return;
    bool valid[NPB];
    int pidx[NPB];
#pragma unroll
    for (int b = 0; b < NPB; ++b) {
        const int pxl = 32 * b + r;
        const int ho = y0 + (pxl >> 4), wo = x0 + (pxl & 15);
        valid[b] = ho < H && wo < W;
        pidx[b] = ho * W + wo;
    }
    if (gn_part) {
        float s1[16], s2[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
#pragma unroll
        for (int b = 0; b < NPB; ++b) {
            if (valid[b]) {
#pragma unroll
                for (int i = 0; i < 16; ++i) { s1[i] += acc[b][i]; s2[i] += acc[b][i] * acc[b][i]; }
            }
        }
#pragma unroll
        for (int m = 16; m > 0; m >>= 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { s1[i] += __shfl_xor(s1[i], m); s2[i] += __shfl_xor(s2[i], m); }
        }
        if (r == 0) {
            float* base = gn_part + (((size_t)n * gridDim.x + blockIdx.x) * 2) * (32 * OB) + 32 * ob + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                *reinterpret_cast<f32x4*>(base + 8 * g) = f32x4{s1[4 * g], s1[4 * g + 1], s1[4 * g + 2], s1[4 * g + 3]};
                *reinterpret_cast<f32x4*>(base + 32 * OB + 8 * g) = f32x4{s2[4 * g], s2[4 * g + 1], s2[4 * g + 2], s2[4 * g + 3]};
            }
        }
    }
    float* on = out + (size_t)n * HW * (32 * OB);
#pragma unroll
    for (int b = 0; b < NPB; ++b) {
        if (valid[b]) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {acc[b][4 * g], acc[b][4 * g + 1], acc[b][4 * g + 2], acc[b][4 * g + 3]};
                *reinterpret_cast<f32x4*>(on + (size_t)pidx[b] * (32 * OB) + 32 * ob + 8 * g + 4 * h) = v;
            }
        }
    }
}

}  // namespace svps

extern "C" int svps_deform_conv_fused_fwd(const float* x_nhwc, const float* offset, const void* wpack, float* out, int N, int C,
                                          int H, int W, int O, int kh, int kw, int pad, int stride, int dil, int Ho, int Wo,
                                          void* stream_) {
    return svps_deform_conv_fused_stats_fwd(x_nhwc, offset, wpack, out, nullptr, N, C, H, W, O, kh, kw, pad, stride, dil, Ho, Wo, stream_);
}

namespace {
// K7'' (the staged form) for the layers it covers: 3 x 3, stride 1, pad 1, dilation 1 (Ho = H, Wo = W); SVPS_K7_HALO=0 keeps the form above
bool dc_use_halo(int C, int H, int W, int Ho, int Wo, int pad, int stride, int dil) {
    static const bool off = [] { const char* e = getenv("SVPS_K7_HALO"); return e && atoi(e) == 0; }();
    return !off && pad == 1 && stride == 1 && dil == 1 && Ho == H && Wo == W && (C % svps::DhLds::CH) == 0;
}
// tile shape of a launch (see DcTile): deterministic in (N, O, Ho Wo) - the GroupNorm partial rows are laid out per tile. Rule from the
// measurements of tools/kbench_k7.py (SVPS_K7_TILE=128 / 256 overrides it for such runs)
bool dc_tile256(int N, int O, int HWo) {
    static const int forced = [] { const char* e = getenv("SVPS_K7_TILE"); return e ? atoi(e) : 0; }();
    if (forced == 128) return false;
    if (forced == 256) return true;
    const long wg128 = (long)((HWo + 127) / 128) * N;
    const int cus = svps_num_cus();
    (void)O;
    return wg128 > cus;     // more than one round of 128-pixel workgroups: the 256-pixel tile wins or ties on every shape of the two towers measured
                            // (T = 5 at 1024 x 2048: 10.10 -> 9.68 ms over the twelve launches; VIPER T = 10: 20.5 -> 18.4); below that it halves the parallelism
}

template <int OB, bool T256>
int dc_launch(const float* x, const float* offset, const void* wpack, float* out, float* gn_partial, int N, int C, int H, int W, int Ho, int Wo,
              int pad, int stride, int dil, hipStream_t stream) {
    using T = svps::DcTile<T256>;
    if (C % T::ch) return SVPS_ERR_BAD_SHAPE;
    auto kern = svps::deform_conv_fused_kernel<OB, T256>;
    static SvpsLdsAttr attr;
    if (hipError_t ae = attr.ensure(reinterpret_cast<const void*>(kern), svps::DcLds<T256>::total); ae != hipSuccess) return (int)ae;
    const int tiles = (Ho * Wo + T::px - 1) / T::px;
    hipLaunchKernelGGL(kern, dim3(tiles, N), dim3(512), svps::DcLds<T256>::total, stream, x, offset, static_cast<const __bf16*>(wpack), out, C, H, W,
                       Ho, Wo, pad, stride, dil, gn_partial);
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int svps_deform_conv_fused_stats_chunks(int N, int O, int Ho, int Wo) {
    if ((O != 128 && O != 256) || Ho <= 0 || Wo <= 0 || N <= 0) return 0;
    // (the statistics entry point is only used by the semantic tower: 3 x 3, stride 1, pad 1 -> the staged form's 8 x 16 tiles)
    if (dc_use_halo(svps::DhLds::CH, Ho, Wo, Ho, Wo, 1, 1, 1)) return ((Ho + svps::DhLds::TH - 1) / svps::DhLds::TH) * ((Wo + svps::DhLds::TW - 1) / svps::DhLds::TW);
    const int px = dc_tile256(N, O, Ho * Wo) ? 256 : 128;
    const int tiles = (Ho * Wo + px - 1) / px;
    return (O == 256 || SVPS_K7_O128_FOUR_WAVES) ? tiles : 2 * tiles;
}

extern "C" int svps_deform_conv_fused_stats_fwd(const float* x_nhwc, const float* offset, const void* wpack, float* out, float* gn_partial,
                                                int N, int C, int H, int W, int O, int kh, int kw, int pad, int stride, int dil, int Ho,
                                                int Wo, void* stream_) {
    if (!x_nhwc || !offset || !wpack || !out) return SVPS_ERR_BAD_ARG;
    if (N <= 0 || H <= 0 || W <= 0 || kh != 3 || kw != 3 || C <= 0 || (C % 64) || (O != 128 && O != 256) || stride <= 0 ||
        dil <= 0 || pad < 0)
        return SVPS_ERR_BAD_SHAPE;
    if (Ho != (H + 2 * pad - (dil * 2 + 1)) / stride + 1 || Wo != (W + 2 * pad - (dil * 2 + 1)) / stride + 1) return SVPS_ERR_BAD_SHAPE;
    if ((size_t)H * W * C >= ((size_t)1 << 31)) return SVPS_ERR_BAD_SHAPE;          // 32-bit pixel indices inside a frame
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const bool t256 = dc_tile256(N, O, Ho * Wo);
    // the partial rows of the statistics are laid out per tile of the form svps_deform_conv_fused_stats_chunks() assumed: the staged
    // form's for a 3 x 3 / stride 1 / pad 1 / dilation 1 layer (the semantic tower's; the only caller with statistics)
    if (gn_partial && dc_use_halo(svps::DhLds::CH, Ho, Wo, Ho, Wo, 1, 1, 1) && !dc_use_halo(C, H, W, Ho, Wo, pad, stride, dil)) return SVPS_ERR_BAD_SHAPE;
    svps_prof_mark(SVPS_KERNEL_DEFORM_CONV, 0, stream);
    int rc;
    if (dc_use_halo(C, H, W, Ho, Wo, pad, stride, dil)) {
        const int tx = (W + svps::DhLds::TW - 1) / svps::DhLds::TW, ty = (H + svps::DhLds::TH - 1) / svps::DhLds::TH;
        hipError_t ae;
        if (O == 256) {
            auto kern = svps::deform_conv_halo_kernel<8>;
            static SvpsLdsAttr attr;
            if ((ae = attr.ensure(reinterpret_cast<const void*>(kern), svps::DhLds::total)) != hipSuccess) return (int)ae;
            hipLaunchKernelGGL(kern, dim3(tx * ty, N), dim3(512), svps::DhLds::total, stream, x_nhwc, offset, static_cast<const __bf16*>(wpack), out,
                               C, H, W, tx, gn_partial);
        } else {
            auto kern = svps::deform_conv_halo_kernel<4>;
            static SvpsLdsAttr attr;
            if ((ae = attr.ensure(reinterpret_cast<const void*>(kern), svps::DhLds::total)) != hipSuccess) return (int)ae;
            hipLaunchKernelGGL(kern, dim3(tx * ty, N), dim3(512), svps::DhLds::total, stream, x_nhwc, offset, static_cast<const __bf16*>(wpack), out,
                               C, H, W, tx, gn_partial);
        }
        rc = (int)hipGetLastError();
        svps_prof_mark(SVPS_KERNEL_DEFORM_CONV, 1, stream);
        return rc;
    }
    if (O == 256) rc = t256 ? dc_launch<8, true>(x_nhwc, offset, wpack, out, gn_partial, N, C, H, W, Ho, Wo, pad, stride, dil, stream)
                            : dc_launch<8, false>(x_nhwc, offset, wpack, out, gn_partial, N, C, H, W, Ho, Wo, pad, stride, dil, stream);
    else rc = t256 ? dc_launch<4, true>(x_nhwc, offset, wpack, out, gn_partial, N, C, H, W, Ho, Wo, pad, stride, dil, stream)
                   : dc_launch<4, false>(x_nhwc, offset, wpack, out, gn_partial, N, C, H, W, Ho, Wo, pad, stride, dil, stream);
    svps_prof_mark(SVPS_KERNEL_DEFORM_CONV, 1, stream);
    return rc;
}
