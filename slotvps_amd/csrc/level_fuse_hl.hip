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
// (hi hi + lo hi + hi lo; the lo lo term is below fp32 resolution). The level maps are stored as TWO fp16 planes [T, HW, 256] (hi, lo):
// 1 KiB per pixel, the bytes of an fp32 map, in the form the consumers' matrix instructions take directly (retr_stats_hl.hip,
// retr_attn.hip and mask_decode.hip have matching forms).
//
// A 1x1 convolution commutes with bilinear interpolation (both are linear, the taps' weights do not depend on the channel), so with
// W = [W_a | W_b] (256 upsampled + 128 incoming channels)
//     f_i = up( W_a f_{i-1} ) + W_b x_i + b
// and the 256-wide part of the product runs at the COARSE resolution (a quarter of the pixels): g_{i-1} = f_{i-1} W_a^T is one launch of
// K8 (slot_gemm.hip, fp16 hi + lo operands) on the fp32 copy this kernel writes for the levels that feed another one; what is left at
// the fine resolution is K = 128: 24 MFMAs per wave and 32-pixel tile instead of 72, 64 weight registers instead of 192 (the first form
// of this kernel - all of K = 384 at the fine resolution, operand tile built from eight scattered tap loads per thread with nothing
// in flight - took 40 ms of a 160-frame step). Level 0: cat(x, x, x) W^T = x (W_1 + W_2 + W_3)^T, the sum formed in float64 on the host.
// The result differs from the reference's order of operations by fp32 rounding only (tests/test_refprec_gpu.py: <= 1e-6 of the map's
// scale against float64).
//
// Mapping: 8 waves, wave w owns output channels [32w, 32w+32) (its 32 x 128 block of W_b as hi and lo A fragments: 64 registers).
// Per 32-pixel tile: all threads build the [32 px][128 ch] operand tile (hi, lo) in LDS from the fp32 NCHW map (loaded one tile ahead);
// every lane requests the 4 taps x 16 channels of g it needs for ITS accumulator entries (pixel = lane & 31) before the MFMAs and
// blends them in fp32 behind them (four tap weights per lane and tile); bias, split, out tiles through LDS so that HBM sees whole
// 512-byte pixel rows of both planes (and of the fp32 copy).
#include <type_traits>

#include "common.h"
#include "../../include/slotvps_hip.h"

namespace svps {

constexpr int kHlK = 128;                        // incoming channels: the contraction at the fine resolution
constexpr int kHlRowBytes = kHlK * 2;            // 256 B per pixel row of an operand tile (16 chunks of 16 B)

struct FuseHlLds {
    static constexpr int a_hi = 0;                                  // [32][128] fp16, 16-B chunks swizzled
    static constexpr int a_lo = kTilePx * kHlRowBytes;
    static constexpr int o_hi = 2 * kTilePx * kHlRowBytes;          // [32][256] fp16 out tiles
    static constexpr int o_lo = o_hi + kTileBytes;
    static constexpr int o_f32 = o_lo + kTileBytes;                 // [32][256] fp32 copy (rows of 1 KiB, 16-B chunks swizzled)
    static constexpr int gtile = o_f32 + 2 * kTileBytes;            // staged taps of g: [2 source rows][18 source columns][1 KiB + 16]
    static constexpr int kGCols = 18, kGRow = 1024 + 16;            // (padded: the lanes of a wave read different columns at the same channel)
    static constexpr int total = gtile + 2 * kGCols * kGRow;
};
static_assert(FuseHlLds::total <= 160 * 1024, "LDS layout");

__device__ __forceinline__ int hl_a_off(int row, int chunk) {       // chunk 0 .. 15
    return row * kHlRowBytes + ((chunk ^ swz(row)) * 16);
}

// x -> (hi, lo) fp16 with hi + lo = x to 22 bits; hi saturates at +-65 504 (an overflow to inf would turn into NaN downstream)
__device__ __forceinline__ void hl_split(float x, _Float16& hi, _Float16& lo) {
    x = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
    asm volatile("" : "+v"(x));          // ONE fp32 value for both halves (hipcc otherwise derives lo from an unrounded product: retr_attn.hip)
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

// STAGED (TAPS, W % 32 == 0: a tile is 32 pixels of ONE output row): the two source rows x <= 18 source columns of g the tile blends arrive
// as whole 1-KiB rows (36 coalesced wave loads per tile instead of 16 loads per LANE that each touch 64 sectors - those were 46 % of
// the kernel), one tile ahead through 20 registers, and every lane reads its four taps from LDS.
// PLANES = false (round 5): only the fp32 result is written - the launches that produce G^(m) = f W_a^m^T for the finer levels (see
// svps_level_fuse_hl_fwd below); the fp32 value is then the unsplit sum itself.
template <bool TAPS, bool F32OUT, bool STAGED = false, bool PLANES = true>
__global__ __launch_bounds__(512) void level_fuse_hl_kernel(
    const float* __restrict__ cur,            // [T, 128, H, W] fp32 (NCHW, the reference's layout)
    const float* __restrict__ gprev,          // TAPS: [T, (H/2)*(W/2), 256] fp32 = f_{i-1} W_a^T, pixel-major
    const _Float16* __restrict__ wb_hi,       // [256, 128] weight of the incoming channels (level 0: W_1 + W_2 + W_3), hi and lo parts
    const _Float16* __restrict__ wb_lo,
    const float* __restrict__ bc,             // [256]
    _Float16* __restrict__ out_hi,            // [T, H*W, 256]
    _Float16* __restrict__ out_lo,
    float* __restrict__ out_f32,              // F32OUT: [T, H*W, 256] the same values as fp32 (operand of the next level's coarse product)
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

    // ---- weight block of this wave: rows 32w .. 32w+31, 8 k-steps, hi and lo
    f16x8 wfh[8], wfl[8];
    {
        const size_t row = (size_t)(32 * w + r) * kHlK + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            wfh[ks] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(wb_hi + row + 16 * ks));
            wfl[ks] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(wb_lo + row + 16 * ks));
        }
    }
    f32x4 bias[4];                            // register 4 g + j <-> channel 32 w + 8 g + 4 h + j
#pragma unroll
    for (int g = 0; g < 4; ++g) bias[g] = *reinterpret_cast<const f32x4*>(bc + 32 * w + 8 * g + 4 * h);

    char* ah = smem + Lds::a_hi;
    char* al = smem + Lds::a_lo;
    // incoming map of one tile: thread = (channel PAIR, 4-pixel group): 2 x 16-B loads (two tiles ahead), 4 four-byte LDS stores per plane
    // (round 5; before: one channel x 8 pixels per thread and 8 TWO-byte stores per plane - twice the LDS instructions and address arithmetic)
    const int ch = 2 * (tid >> 3), pg = tid & 7;
    const float* src = cur + ((size_t)t * 128 + ch) * HW;
    const bool aligned = (HW & 3) == 0;
    // TWO register sets (round 5): the incoming map (and the rows of g) are requested TWO tiles ahead - HBM latency under load is longer
    // than one tile period of this lock-step workgroup (one tile ahead left every commit waiting for its loads)
    f32x4 c0[2], c1[2];
    auto fetch = [&](int tile, auto par) {
        constexpr int P = decltype(par)::value;
        const int pp = px_begin + tile * kTilePx + 4 * pg;                 // c0: channel ch, c1: channel ch + 1, pixels pp .. pp + 3
        if (pp + 4 <= HW && aligned) {
            c0[P] = *reinterpret_cast<const f32x4*>(src + pp);
            c1[P] = *reinterpret_cast<const f32x4*>(src + HW + pp);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                c0[P][j] = src[pp + j < HW ? pp + j : HW - 1];
                c1[P][j] = src[HW + (pp + j < HW ? pp + j : HW - 1)];
            }
        }
    };
    auto commit = [&](auto par) {
        constexpr int P = decltype(par)::value;
        typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = 4 * pg + j;
            f16x2_t vh, vl;
            _Float16 h0_, l0_, h1_, l1_;
            hl_split(c0[P][j], h0_, l0_);
            hl_split(c1[P][j], h1_, l1_);
            vh[0] = h0_; vh[1] = h1_; vl[0] = l0_; vl[1] = l1_;
            const int o = hl_a_off(row, ch >> 3) + (ch & 7) * 2;             // (ch even: a 4-byte slot)
            *reinterpret_cast<f16x2_t*>(ah + o) = vh;
            *reinterpret_cast<f16x2_t*>(al + o) = vl;
        }
    };
    // out tiles -> HBM, all 512 threads: whole 512-byte (fp32: 1-KiB) pixel rows
    auto store_out = [&](int tile) {
#pragma unroll
        for (int u = 0; u < (PLANES ? 2 : 0); ++u) {
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
        if constexpr (F32OUT) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int piece = u * 512 + tid;                        // [row][64 chunk positions of 16 B]
                const int row = piece >> 6, cpos = piece & 63;
                const int px = px_begin + tile * kTilePx + row;
                const u32x4 v = *reinterpret_cast<const u32x4*>(smem + Lds::o_f32 + row * 1024 + cpos * 16);
                if (px < px_end) *reinterpret_cast<u32x4*>(out_f32 + ((size_t)t * HW + px) * kD + ((cpos ^ (row & 15)) * 4)) = v;
            }
        }
    };

    // STAGED: wave w stages items w, w + 8, .. (< 36) of the tile's [2 rows][18 columns] of g: one 1-KiB row per wave instruction
    f32x4 gt[2][5];
    auto fetch_g = [&](int tile, auto par) {
        constexpr int P = decltype(par)::value;
        const int px0 = px_begin + tile * kTilePx;         // first pixel of the tile: x0 % 32 == 0, one output row
        const int y = px0 / W, x0 = px0 - y * W;
        const float sy = fmaxf((y + 0.5f) * 0.5f - 0.5f, 0.f);
        const int y0 = (int)sy, y1 = y0 + 1 < Hp ? y0 + 1 : Hp - 1;
        const int c_lo = x0 / 2 - 1 > 0 ? x0 / 2 - 1 : 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int item = w + 8 * i;                    // 0 .. 35 (row = item / 18, column = item % 18); wave-uniform
            if (item < 2 * Lds::kGCols) {
                const int row = item / Lds::kGCols;
                int col = c_lo + (item - row * Lds::kGCols);
                col = col < Wp ? col : Wp - 1;
                gt[P][i] = *reinterpret_cast<const f32x4*>(gprev + ((size_t)t * Hp * Wp + (size_t)(row ? y1 : y0) * Wp + col) * kD + 4 * lane);
            }
        }
    };
    auto commit_g = [&](auto par) {
        constexpr int P = decltype(par)::value;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int item = w + 8 * i;
            if (item < 2 * Lds::kGCols) *reinterpret_cast<f32x4*>(smem + Lds::gtile + item * Lds::kGRow + 16 * lane) = gt[P][i];
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    fetch(0, I0{});
    if constexpr (STAGED) fetch_g(0, I0{});
    if (nt > 1) {
        fetch(1, I1{});
        if constexpr (STAGED) fetch_g(1, I1{});
    }
    commit(I0{});
    if constexpr (STAGED) commit_g(I0{});
    // One tile. Invariant at the top: LDS holds the operand tile (and the rows of g) of tile `it`; register set P ^ 1 holds tile it + 1
    // (requested one whole iteration ago); set P is free and takes tile it + 2.
    auto body = [&](int it, auto par) {
        constexpr int P = decltype(par)::value;
        using IP = std::integral_constant<int, P>;
        using IQ = std::integral_constant<int, P ^ 1>;
        __syncthreads();                                   // operand tile `it` complete; out tiles of it-1 have been read
        if (it + 2 < nt) {
            fetch(it + 2, IP{});
            if constexpr (STAGED) fetch_g(it + 2, IP{});
        }
        // ---- per-lane taps of g (non-staged form: this lane's accumulator entries - pixel r, channels 32 w + 8 g + 4 h + j - requested
        // before the MFMAs)
        f32x4 tp[4][4];                                    // [tap][g]
        float h1 = 0.f, w1 = 0.f;
        const char* a00 = nullptr;
        const char* a01 = nullptr;
        if constexpr (TAPS && STAGED) {
            const int px0 = px_begin + it * kTilePx;
            const int y = px0 / W, x = px0 - y * W + r;
            const float sy = fmaxf((y + 0.5f) * 0.5f - 0.5f, 0.f), sx = fmaxf((x + 0.5f) * 0.5f - 0.5f, 0.f);
            const int y0 = (int)sy, xs0 = (int)sx;
            const int xs1 = xs0 + 1 < Wp ? xs0 + 1 : Wp - 1;
            h1 = sy - (float)y0;
            w1 = sx - (float)xs0;
            const int c_lo = (px0 - y * W) / 2 - 1 > 0 ? (px0 - y * W) / 2 - 1 : 0;
            const char* g0 = smem + Lds::gtile + (32 * w + 4 * h) * 4;
            a00 = g0 + (xs0 - c_lo) * Lds::kGRow;
            a01 = g0 + (xs1 - c_lo) * Lds::kGRow;          // (the taps themselves are read from LDS group by group behind the MFMAs)
        } else if constexpr (TAPS) {
            int pp = px_begin + it * kTilePx + r;
            pp = pp < HW ? pp : HW - 1;
            const int y = pp / W, x = pp - y * W;
            // F.interpolate(scale 2, bilinear, align_corners=False): source coordinate (d + 0.5) / 2 - 0.5 clamped at 0
            const float sy = fmaxf((y + 0.5f) * 0.5f - 0.5f, 0.f), sx = fmaxf((x + 0.5f) * 0.5f - 0.5f, 0.f);
            const int y0 = (int)sy, x0 = (int)sx;
            const int y1 = y0 + 1 < Hp ? y0 + 1 : Hp - 1, x1 = x0 + 1 < Wp ? x0 + 1 : Wp - 1;
            h1 = sy - (float)y0;
            w1 = sx - (float)x0;
            const float* gb = gprev + (size_t)t * Hp * Wp * kD + 32 * w + 4 * h;
            const float* t00 = gb + ((size_t)y0 * Wp + x0) * kD;
            const float* t01 = gb + ((size_t)y0 * Wp + x1) * kD;
            const float* t10 = gb + ((size_t)y1 * Wp + x0) * kD;
            const float* t11 = gb + ((size_t)y1 * Wp + x1) * kD;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                tp[0][g] = *reinterpret_cast<const f32x4*>(t00 + 8 * g);
                tp[1][g] = *reinterpret_cast<const f32x4*>(t01 + 8 * g);
                tp[2][g] = *reinterpret_cast<const f32x4*>(t10 + 8 * g);
                tp[3][g] = *reinterpret_cast<const f32x4*>(t11 + 8 * g);
            }
        }
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[4 * g + j] = bias[g][j];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const f16x8 xh = *reinterpret_cast<const f16x8*>(ah + hl_a_off(r, 2 * ks + h));
            const f16x8 xl = *reinterpret_cast<const f16x8*>(al + hl_a_off(r, 2 * ks + h));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfl[ks], xh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfh[ks], xl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfh[ks], xh, acc, 0, 0, 0);
        }
        // ---- + up(g) in fp32: w00 a + w01 b + w10 c + w11 d with the four tap weights formed once per lane and tile (round 5: four
        // instead of seven vector instructions per element; torch's upsample_bilinear2d groups the same sum as (1-ly) ((1-lx) a + lx b) +
        // ly ((1-lx) c + lx d) - an fp32 rounding of difference, inside the 5e-6 bound against float64 of tests/test_refprec_gpu.py); split; out tiles
        const float h0 = 1.f - h1, w0 = 1.f - w1;
        const float w00 = h0 * w0, w01 = h0 * w1, w10 = h1 * w0, w11 = h1 * w1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch0 = 32 * w + 8 * g + 4 * h;
            if constexpr (TAPS && STAGED) {
                tp[0][g] = *reinterpret_cast<const f32x4*>(a00 + 32 * g);
                tp[1][g] = *reinterpret_cast<const f32x4*>(a01 + 32 * g);
                tp[2][g] = *reinterpret_cast<const f32x4*>(a00 + Lds::kGCols * Lds::kGRow + 32 * g);
                tp[3][g] = *reinterpret_cast<const f32x4*>(a01 + Lds::kGCols * Lds::kGRow + 32 * g);
            }
            f16x4 oh, ol;
            f32x4 of;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = acc[4 * g + j];
                if constexpr (TAPS) v = fmaf(w11, tp[3][g][j], fmaf(w10, tp[2][g][j], fmaf(w01, tp[1][g][j], fmaf(w00, tp[0][g][j], v))));
                if constexpr (PLANES) {
                    _Float16 vh, vl;
                    hl_split(v, vh, vl);
                    oh[j] = vh;
                    ol[j] = vl;
                    of[j] = (float)vh + (float)vl;         // the fp32 copy holds exactly the planes' value
                } else {
                    of[j] = v;
                }
            }
            const int o = r * kRowBytes + (((ch0 >> 3) ^ swz(r)) * 16) + (ch0 & 7) * 2;
            if constexpr (PLANES) {
                *reinterpret_cast<f16x4*>(smem + Lds::o_hi + o) = oh;
                *reinterpret_cast<f16x4*>(smem + Lds::o_lo + o) = ol;
            }
            if constexpr (F32OUT) *reinterpret_cast<f32x4*>(smem + Lds::o_f32 + r * 1024 + (((ch0 >> 2) ^ (r & 15)) * 16)) = of;
        }
        __syncthreads();                                   // out tiles complete; every wave is done reading operand tile `it` (and its taps)
        if (it + 1 < nt) {
            commit(IQ{});                                  // operand tile it + 1: its loads were issued a whole iteration ago
            if constexpr (STAGED) commit_g(IQ{});
        }
        store_out(it);
    };
    for (int it = 0; it < nt; it += 2) {
        body(it, I0{});
        if (it + 1 < nt) body(it + 1, I1{});
    }
}

}  // namespace svps

// svps_level_fuse_hl_fwd (include/slotvps_hip.h): out = up(gprev) + wb cur + bc. cur [T, 128, H, W] fp32 NCHW; gprev [T, (H/2)(W/2), 256]
// fp32 or NULL (no upsampled term); wb_hi / wb_lo [256, 128] fp16; bc [256] fp32; out_hi / out_lo [T, H*W, 256] fp16 planes (both or
// neither); out_f32 the same values as fp32 [T, H*W, 256] or NULL (at least one output).
// The level recursion without any 256-wide product (round 5): with G^(m)_i = f_i (W_a^m)^T,
//     G^(m)_i = up( G^(m+1)_{i-1} ) + (W_a^m W_b) x_i + W_a^m b          (level 0: W_a^m (W_1 + W_2 + W_3) x_0 + W_a^m b)
// - the 1x1 conv commutes with the interpolation at EVERY level, so the host composes W_a^m W_b in float64 once per weight version and
// level i is 4 - i launches of this kernel (m = 0: the planes of f_i; m >= 1: fp32 only), each a K = 128 product at level i's resolution.
extern "C" int svps_level_fuse_hl_fwd(const float* cur, const float* gprev, const void* wb_hi, const void* wb_lo, const float* bc,
                                      void* out_hi, void* out_lo, float* out_f32, int T, int H, int W, void* stream_) {
    if (!cur || !wb_hi || !wb_lo || !bc || (!out_hi != !out_lo) || (!out_hi && !out_f32)) return SVPS_ERR_BAD_ARG;
    if (T <= 0 || H <= 0 || W <= 0) return SVPS_ERR_BAD_SHAPE;
    if ((size_t)H * W > svps::kMaxFramePixels) return SVPS_ERR_BAD_SHAPE;
    if (gprev && ((H & 1) || (W & 1))) return SVPS_ERR_BAD_SHAPE;     // x2 upsampling: even sizes
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    using H16 = _Float16;
    const int HW = H * W;
    const int tiles = (HW + svps::kTilePx - 1) / svps::kTilePx;
    int chunks = svps_pick_chunks(T, tiles, svps_num_cus());
    const int tpc = (tiles + chunks - 1) / chunks;
    chunks = (tiles + tpc - 1) / tpc;
    constexpr int lds = svps::FuseHlLds::total;
    svps_prof_mark(SVPS_KERNEL_LEVEL_FUSE, 0, stream);
    hipError_t e = hipSuccess;
#define SVPS_LFH(TAPS, F32, PL)                                                                                                    \
    do {                                                                                                                           \
        const bool staged = TAPS && (W & 31) == 0;          /* a tile = 32 pixels of one output row: taps through LDS */           \
        auto kern = staged ? svps::level_fuse_hl_kernel<TAPS, F32, TAPS, PL> : svps::level_fuse_hl_kernel<TAPS, F32, false, PL>;  \
        static SvpsLdsAttr attr[2];                                                                                                \
        if ((e = attr[staged ? 1 : 0].ensure(reinterpret_cast<const void*>(kern), lds)) != hipSuccess) return (int)e;             \
        hipLaunchKernelGGL(kern, dim3(chunks, T), dim3(512), lds, stream, cur, gprev, static_cast<const H16*>(wb_hi),              \
                           static_cast<const H16*>(wb_lo), bc, static_cast<H16*>(out_hi), static_cast<H16*>(out_lo), out_f32, H, W, tpc); \
    } while (0)
    if (!out_hi) { if (gprev) SVPS_LFH(true, true, false); else SVPS_LFH(false, true, false); }
    else if (gprev) { if (out_f32) SVPS_LFH(true, true, true); else SVPS_LFH(true, false, true); }
    else { if (out_f32) SVPS_LFH(false, true, true); else SVPS_LFH(false, false, true); }
#undef SVPS_LFH
    e = hipGetLastError();
    svps_prof_mark(SVPS_KERNEL_LEVEL_FUSE, 1, stream);
    return (int)e;
}
