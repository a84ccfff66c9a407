// K4-HL - level fusion of the multi-scale slot head at the REFERENCE's precision on the matrix cores (round 4).
//
// Same function as level_fuse.hip (MultiScaleDynamicMaskHead.forward, mmdet/models/detectors/dynamic_mask_head.py:171-188):
//
//     level i > 0:  f_i = conv1x1_{384->256}( cat( bilinear_x2(f_{i-1}), x_i ) )        :178-181
//     level 0:      f_0 = conv1x1_{384->256}( cat( x_0, x_0, x_0 ) )                    :182-185
//
// but nothing is rounded to 16 bits as a VALUE. The reference runs this path in fp32 (vps_temporal_slots.py:55); the fp32 matrix
// instructions of gfx950 run at the vector rate, so every 16-bit matrix operand is carried as FP16 hi + lo instead (hi = fp16(x),
// lo = fp16(x - hi): 22 bits of mantissa, |x| < 65 504) and a product takes three MFMAs into one fp32 accumulator
// (hi hi + lo hi + hi lo; the lo lo term is below fp32 resolution):
//     * the level maps are stored as TWO fp16 planes [T, HW, 256] (hi, lo) - 1 KiB per pixel, the bytes of an fp32 map, in the form
//       the consumers' matrix instructions take directly (retr_stats_t.hip, retr_attn.hip and mask_decode.hip have matching forms)
//     * the incoming fp32 NCHW map, the fp32 bilinear blend of the previous level's (hi + lo) taps and the conv weight are split the same way
//     * accumulation, bias and the blend itself (torch's upsample_bilinear2d expression) are fp32.
// Against a float64 evaluation of the reference's formulas: <= 2e-6 of the map's scale (tests/test_refprec_gpu.py); the exact mode's
// fp32 vector-ALU kernel (exact_f32.hip) measures the same.
//
// Mapping: 8 waves, wave w owns output channels [32w, 32w+32): its 32 x 384 weight block as hi AND lo A fragments stays in 192
// registers. A plain kernel: per 32-pixel tile the workgroup builds the [32 px][384 ch] operand tile (hi and lo) in LDS, runs 72 MFMAs
// per wave, splits the result and sends both out tiles through LDS so that HBM sees whole 512-byte pixel rows. No register prefetch
// across the matrix phase (the weights leave no room for it): bound by its serial phases, ~3x the time of the 16-bit kernel for 3x
// its matrix work and twice its output bytes.
#include "common.h"
#include "../../include/slotvps_hip.h"

namespace svps {

constexpr int kHlIn = 384;
constexpr int kHlRowBytes = kHlIn * 2;            // 768 B per pixel row of an operand tile

struct FuseHlLds {
    static constexpr int a_hi = 0;                                  // [32][384] fp16, 16-B chunks swizzled
    static constexpr int a_lo = kTilePx * kHlRowBytes;
    static constexpr int o_hi = 2 * kTilePx * kHlRowBytes;          // [32][256] fp16 out tiles
    static constexpr int o_lo = o_hi + kTileBytes;
    static constexpr int total = o_lo + kTileBytes;
};

__device__ __forceinline__ int hl_a_off(int row, int chunk) {       // level_fuse.hip's operand-tile swizzle
    return row * kHlRowBytes + (((chunk & ~15) | ((chunk ^ swz(row)) & 15)) * 16);
}

// x -> (hi, lo) fp16 with hi + lo = x to 22 bits; hi saturates at +-65 504 (an overflow to inf would turn into NaN downstream)
__device__ __forceinline__ void hl_split(float x, _Float16& hi, _Float16& lo) {
    x = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
    asm volatile("" : "+v"(x));          // ONE fp32 value for both halves (hipcc otherwise derives lo from an unrounded product: retr_attn.hip)
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

template <bool LEVEL0>
__global__ __launch_bounds__(512) void level_fuse_hl_kernel(
    const float* __restrict__ cur,            // [T, 128, H, W] fp32 (NCHW, the reference's layout)
    const _Float16* __restrict__ prev_hi,     // [T, (H/2)*(W/2), 256] pixel-major (unused for LEVEL0)
    const _Float16* __restrict__ prev_lo,
    const _Float16* __restrict__ wc_hi,       // [256, 384] conv weight (row = output channel), hi and lo parts
    const _Float16* __restrict__ wc_lo,
    const float* __restrict__ bc,             // [256]
    _Float16* __restrict__ out_hi,            // [T, H*W, 256]
    _Float16* __restrict__ out_lo,
    int H, int W, int tiles_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using Lds = FuseHlLds;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int t = blockIdx.y, c = blockIdx.x;
    const int HW = H * W;
    const int Hp = H >> 1, Wp = W >> 1;

    const int px_begin = c * tiles_per_chunk * kTilePx;
    int px_end = px_begin + tiles_per_chunk * kTilePx;
    px_end = px_end < HW ? px_end : HW;
    const int nt = (px_end - px_begin + kTilePx - 1) / kTilePx;

    // ---- weight block of this wave: rows 32w .. 32w+31, 24 k-steps, hi and lo ---------------------------------
    f16x8 wfh[24], wfl[24];
    {
        const size_t row = (size_t)(32 * w + r) * kHlIn + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 24; ++ks) {
            wfh[ks] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(wc_hi + row + 16 * ks));
            wfl[ks] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(wc_lo + row + 16 * ks));
        }
    }

    char* ah = smem + Lds::a_hi;
    char* al = smem + Lds::a_lo;
    // ---- operand tile of one 32-pixel tile (hi and lo), all 512 threads --------------------------------------
    auto build = [&](int tile) {
        const int px0 = px_begin + tile * kTilePx;
        {   // (a) incoming 128-channel map -> chunks 32..47 (channels 256..383); LEVEL0 also -> 0..15, 16..31
            //     thread = (channel, 8-pixel group): 2 x 16-B loads, 8 two-byte LDS stores per plane
            const int ch = tid >> 2, pg = tid & 3;
            const int pp = px0 + 8 * pg;
            const float* src = cur + ((size_t)t * 128 + ch) * HW;
            float v[8];
            if (pp + 8 <= HW && (HW & 3) == 0) {
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(src + pp);
                const f32x4 c1 = *reinterpret_cast<const f32x4*>(src + pp + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = c0[j]; v[4 + j] = c1[j]; }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = src[pp + j < HW ? pp + j : HW - 1];
            }
            const int chunk = 32 + (ch >> 3);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = 8 * pg + j;
                _Float16 vh, vl;
                hl_split(v[j], vh, vl);
                const int o = hl_a_off(row, chunk) + (ch & 7) * 2;
                *reinterpret_cast<_Float16*>(ah + o) = vh;
                *reinterpret_cast<_Float16*>(al + o) = vl;
                if constexpr (LEVEL0) {
                    const int o1 = hl_a_off(row, chunk - 32) + (ch & 7) * 2, o2 = hl_a_off(row, chunk - 16) + (ch & 7) * 2;
                    *reinterpret_cast<_Float16*>(ah + o1) = vh;
                    *reinterpret_cast<_Float16*>(al + o1) = vl;
                    *reinterpret_cast<_Float16*>(ah + o2) = vh;
                    *reinterpret_cast<_Float16*>(al + o2) = vl;
                }
            }
        }
        if constexpr (!LEVEL0) {
            // (b) upsampled previous level -> chunks 0..31: thread = (pixel, chunks ck and ck + 16), four taps each, one chunk at a time.
            //     F.interpolate(scale 2, bilinear, align_corners=False): source coordinate (d + 0.5) / 2 - 0.5 clamped at 0; torch's
            //     upsample_bilinear2d expression (1-ly) ((1-lx) a + lx b) + ly ((1-lx) c + lx d) in fp32 on the exact tap values hi + lo
            const int px = tid >> 4, ck = tid & 15;
            int pp = px0 + px;
            pp = pp < HW ? pp : HW - 1;
            const int y = pp / W, x = pp - y * W;
            const float sy = fmaxf((y + 0.5f) * 0.5f - 0.5f, 0.f), sx = fmaxf((x + 0.5f) * 0.5f - 0.5f, 0.f);
            const int y0 = (int)sy, x0 = (int)sx;
            const int y1 = y0 + 1 < Hp ? y0 + 1 : Hp - 1, x1 = x0 + 1 < Wp ? x0 + 1 : Wp - 1;
            const float h1 = sy - (float)y0, w1 = sx - (float)x0;
            const float h0 = 1.f - h1, w0 = 1.f - w1;
            const size_t fb = (size_t)t * Hp * Wp * kD;
            const size_t o00 = fb + ((size_t)y0 * Wp + x0) * kD, o01 = fb + ((size_t)y0 * Wp + x1) * kD;
            const size_t o10 = fb + ((size_t)y1 * Wp + x0) * kD, o11 = fb + ((size_t)y1 * Wp + x1) * kD;
#pragma unroll 1
            for (int q = 0; q < 4; ++q) {                              // (chunk u = q >> 1, half of its 8 channels): 4 channels per step -
                const int u = q >> 1, hf = q & 1;                      // the 192 weight registers leave room for little else
                const int co = 8 * (ck + 16 * u) + 4 * hf;
                auto tap = [&](size_t o, f32x4& dst) {
                    const f16x4 th = *reinterpret_cast<const f16x4*>(prev_hi + o + co);
                    const f16x4 tl = *reinterpret_cast<const f16x4*>(prev_lo + o + co);
#pragma unroll
                    for (int j = 0; j < 4; ++j) dst[j] = (float)th[j] + (float)tl[j];      // exact: 22 bits
                };
                f32x4 a, b, cc, d;
                tap(o00, a); tap(o01, b); tap(o10, cc); tap(o11, d);
                f16x4 oh, ol;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = h0 * (w0 * a[j] + w1 * b[j]) + h1 * (w0 * cc[j] + w1 * d[j]);
                    _Float16 vh, vl;
                    hl_split(v, vh, vl);
                    oh[j] = vh;
                    ol[j] = vl;
                }
                *reinterpret_cast<f16x4*>(ah + hl_a_off(px, ck + 16 * u) + 8 * hf) = oh;
                *reinterpret_cast<f16x4*>(al + hl_a_off(px, ck + 16 * u) + 8 * hf) = ol;
            }
        }
    };
    // out tiles -> HBM, all 512 threads: 2 x 16 KiB per tile, 2 x 2 x 16 B per thread, whole 512-byte pixel rows
    auto store_out = [&](int tile) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int piece = u * 512 + tid;                            // [row][chunk position]
            const int row = piece >> 5, cpos = piece & 31;
            const int px = px_begin + tile * kTilePx + row;
            const u32x4 vh = *reinterpret_cast<const u32x4*>(smem + Lds::o_hi + row * kRowBytes + cpos * 16);
            const u32x4 vl = *reinterpret_cast<const u32x4*>(smem + Lds::o_lo + row * kRowBytes + cpos * 16);
            if (px < px_end) {
                const size_t o = ((size_t)t * HW + px) * kD + ((cpos ^ swz(row)) * 8);
                *reinterpret_cast<u32x4*>(out_hi + o) = vh;
                *reinterpret_cast<u32x4*>(out_lo + o) = vl;
            }
        }
    };

    for (int it = 0; it < nt; ++it) {
        build(it);
        __syncthreads();                                   // operand tile `it` complete; out tiles of it-1 have been read (barrier below)
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 24; ++ks) {
            const f16x8 xh = *reinterpret_cast<const f16x8*>(ah + hl_a_off(r, 2 * ks + h));
            const f16x8 xl = *reinterpret_cast<const f16x8*>(al + hl_a_off(r, 2 * ks + h));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfl[ks], xh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfh[ks], xl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfh[ks], xh, acc, 0, 0, 0);
        }
        // split the result (+ bias) and write this wave's 32 channels of both out tiles: register 4 g + j <-> channel 32 w + 8 g + 4 h + j
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch0 = 32 * w + 8 * g + 4 * h;
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bc + ch0);
            f16x4 oh, ol;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                _Float16 vh, vl;
                hl_split(acc[4 * g + j] + b4[j], vh, vl);
                oh[j] = vh;
                ol[j] = vl;
            }
            const int o = r * kRowBytes + (((ch0 >> 3) ^ swz(r)) * 16) + (ch0 & 7) * 2;
            *reinterpret_cast<f16x4*>(smem + Lds::o_hi + o) = oh;
            *reinterpret_cast<f16x4*>(smem + Lds::o_lo + o) = ol;
        }
        __syncthreads();                                   // out tiles complete; every wave is done reading operand tile `it`
        store_out(it);
    }
}

}  // namespace svps

// svps_level_fuse_hl_fwd (include/slotvps_hip.h): cur [T, 128, H, W] fp32 NCHW; prev_hi / prev_lo [T, (H/2)(W/2), 256] fp16 or both NULL
// (level 0); wc_hi / wc_lo [256, 384] fp16; bc [256] fp32; out_hi / out_lo [T, H*W, 256] fp16.
extern "C" int svps_level_fuse_hl_fwd(const float* cur, const void* prev_hi, const void* prev_lo, const void* wc_hi, const void* wc_lo,
                                      const float* bc, void* out_hi, void* out_lo, int T, int H, int W, void* stream_) {
    if (!cur || !wc_hi || !wc_lo || !bc || !out_hi || !out_lo || ((prev_hi == nullptr) != (prev_lo == nullptr))) return SVPS_ERR_BAD_ARG;
    if (T <= 0 || H <= 0 || W <= 0) return SVPS_ERR_BAD_SHAPE;
    if ((size_t)H * W > svps::kMaxFramePixels) return SVPS_ERR_BAD_SHAPE;
    if (prev_hi && ((H & 1) || (W & 1))) return SVPS_ERR_BAD_SHAPE;   // x2 upsampling: even sizes
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    using H16 = _Float16;
    const int HW = H * W;
    const int tiles = (HW + svps::kTilePx - 1) / svps::kTilePx;
    int chunks = svps_pick_chunks(T, tiles, svps_num_cus());
    const int tpc = (tiles + chunks - 1) / chunks;
    chunks = (tiles + tpc - 1) / tpc;
    constexpr int lds = svps::FuseHlLds::total;
    svps_prof_mark(SVPS_KERNEL_LEVEL_FUSE, 0, stream);
    hipError_t e;
    if (prev_hi) {
        auto kern = svps::level_fuse_hl_kernel<false>;
        static SvpsLdsAttr attr;
        if ((e = attr.ensure(reinterpret_cast<const void*>(kern), lds)) != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, dim3(chunks, T), dim3(512), lds, stream, cur, static_cast<const H16*>(prev_hi), static_cast<const H16*>(prev_lo),
                           static_cast<const H16*>(wc_hi), static_cast<const H16*>(wc_lo), bc, static_cast<H16*>(out_hi), static_cast<H16*>(out_lo), H, W, tpc);
    } else {
        auto kern = svps::level_fuse_hl_kernel<true>;
        static SvpsLdsAttr attr;
        if ((e = attr.ensure(reinterpret_cast<const void*>(kern), lds)) != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, dim3(chunks, T), dim3(512), lds, stream, cur, (const H16*)nullptr, (const H16*)nullptr,
                           static_cast<const H16*>(wc_hi), static_cast<const H16*>(wc_lo), bc, static_cast<H16*>(out_hi), static_cast<H16*>(out_lo), H, W, tpc);
    }
    e = hipGetLastError();
    svps_prof_mark(SVPS_KERNEL_LEVEL_FUSE, 1, stream);
    return (int)e;
}
