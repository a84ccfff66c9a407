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

#ifdef SVPS_K4HL_STAMP
// diagnostic build only (tools/k4hl_stamps.py): s_memtime stamps of waves 0 and 4 of one workgroup, tiles 8 .. 15
__device__ unsigned long long k4hl_stamps[2][8][8];          // [wave 0 / 4][tile - 8][point]
#define K4HL_STAMP(pt)                                                                                              \
    do {                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        if (blockIdx.x == 3 && blockIdx.y == 2 && (w & 3) == 0 && it >= 8 && it < 16 && lane == 0)                  \
            k4hl_stamps[w >> 2][it - 8][pt] = __builtin_amdgcn_s_memtime();                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
    } while (0)
#else
#define K4HL_STAMP(pt) do {} while (0)
#endif

constexpr int kHlK = 128;                        // incoming channels: the contraction at the fine resolution
constexpr int kHlRowBytes = kHlK * 2;            // 256 B per pixel row of an operand tile (16 chunks of 16 B)

struct FuseHlLds {
    static constexpr int a_hi = 0;                                  // [32][128] fp16, 16-B chunks swizzled
    static constexpr int a_lo = kTilePx * kHlRowBytes;
    static constexpr int o_hi = 2 * kTilePx * kHlRowBytes;          // [32][256] fp16 out tiles
    static constexpr int o_lo = o_hi + kTileBytes;
    static constexpr int o_f32 = o_lo + kTileBytes;                 // [32][256] fp32 copy (rows of 1 KiB, 16-B chunks swizzled)
    static constexpr int gtile = o_f32 + 2 * kTileBytes;            // staged taps of g: [3 source rows (row % 3)][18 source columns][1 KiB + 16]
    static constexpr int kGCols = 18, kGRow = 1024 + 16;            // (padded: the lanes of a wave read different columns at the same channel)
    static constexpr int kGRows = 3;                                // a ring: walking down a column strip, two output rows share their source rows
    static constexpr int total = gtile + kGRows * kGCols * kGRow;
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

// The same for a PAIR of values (round 5): on gfx950 the vector forms compile to packed instructions - v_cvt_pk_f16_f32, v_pk_add_f32 (and
// v_pk_fma_f32 for the blend below) - two values per issue slot; this kernel is bound by vector issue (12 VALU per MFMA before).
typedef __attribute__((ext_vector_type(2))) float hl_f32x2;
typedef __attribute__((ext_vector_type(2))) _Float16 hl_f16x2;
__device__ __forceinline__ void hl_split2(hl_f32x2 x, hl_f16x2& hi, hl_f16x2& lo) {
    x[0] = __builtin_amdgcn_fmed3f(x[0], -65504.f, 65504.f);
    x[1] = __builtin_amdgcn_fmed3f(x[1], -65504.f, 65504.f);
    asm volatile("" : "+v"(x));          // ONE fp32 pair for both halves (see hl_split)
    hi = __builtin_convertvector(x, hl_f16x2);
    lo = __builtin_convertvector(x - __builtin_convertvector(hi, hl_f32x2), hl_f16x2);
}

// STAGED (TAPS, W % 32 == 0: a tile is 32 pixels of ONE output row): the two source rows x <= 18 source columns of g the tile blends arrive
// as whole 1-KiB rows (coalesced wave loads instead of 16 loads per LANE that each touch 64 sectors - those were 46 % of the kernel),
// two tiles ahead through 20 registers, and every lane reads its four taps from LDS. A workgroup walks DOWN a 32-pixel column strip
// (round 5; tile sequence number s -> strip s / H, output row s % H): two consecutive output rows blend the same two source rows and
// the next pair needs ONE new row, so the rows live in a ring of three and a tile stages 18 rows of 1 KiB every other tile instead of
// 36 every tile (s_memtime stamps, tools/k4hl_stamps.py: issuing those loads behind the previous tile's stores was 1 000 - 3 000 of
// a tile's 7 500 cycles, and the younger waves held everyone at the second barrier).
// PLANES = false (round 5): only the fp32 result is written - the launches that produce G^(m) = f W_a^m^T for the finer levels (see
// svps_level_fuse_hl_fwd below); the fp32 value is then the unsplit sum itself.
// (strip, row) of a tile sequence number of the column-strip walk, advanced one tile at a time: every consumer of a tile's position
// (requests, commits, taps, stores) keeps its own cursor instead of dividing the sequence number by H - this kernel is bound by the
// instructions it issues, not by memory (timing-only ablation of round 5: 72 % of its time remains with NO global traffic at all)
struct HlCursor {
    int strip, y;
    __device__ __forceinline__ void next(int H) { if (++y == H) { y = 0; ++strip; } }
    __device__ __forceinline__ int px0(int W) const { return y * W + 32 * strip; }
};

// PMIN (round 6): the incoming map arrives as fp16 hi + lo PIXEL-MAJOR planes [T, HW, 128] (cur_ = hi, cur_lo_ = lo: the semantic tower's
// own rows, gn_relu.hip) - a thread moves one 16-byte chunk of each plane straight into the operand tile: no transposition, no split.
template <bool TAPS, bool F32OUT, bool STAGED = false, bool PLANES = true, bool PMIN = false>
__device__ __forceinline__ void level_fuse_hl_body(
    const int t, const int c,                 // frame and chunk of this workgroup (level_fuse_hl_kernel: blockIdx.y, .x; the multi-order kernel decodes them)
    const void* __restrict__ cur_,            // [T, 128, H, W] fp32 (NCHW, the reference's layout); PMIN: [T, HW, 128] fp16, the hi plane
    const void* __restrict__ cur_lo_,         // PMIN: the lo plane
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
    const float* cur = static_cast<const float*>(cur_);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int HW = H * W;
    const int Hp = H >> 1, Wp = W >> 1;

    const int px_begin = c * tiles_per_chunk * kTilePx;            // row-major forms: the chunk is a run of pixels
    int px_end = px_begin + tiles_per_chunk * kTilePx;
    px_end = px_end < HW ? px_end : HW;
    int nt = (px_end - px_begin + kTilePx - 1) / kTilePx;
    const int s_begin = c * tiles_per_chunk;                       // STAGED: a run of tile sequence numbers (column strips, all tiles full)
    if constexpr (STAGED) {
        px_end = HW;
        const int left = H * (W >> 5) - s_begin;
        nt = left < tiles_per_chunk ? left : tiles_per_chunk;
    }
    HlCursor c_first = {0, 0};                                     // the workgroup's first tile
    if constexpr (STAGED) {
        c_first.strip = s_begin / H;
        c_first.y = s_begin - c_first.strip * H;
    }
    // first pixel of the tile a consumer's cursor points at; the cursor moves on (row-major forms: from the tile number)
    auto take_px0 = [&](HlCursor& cu, int tile) {
        if constexpr (STAGED) {
            const int p = cu.px0(W);
            cu.next(H);
            return p;
        } else {
            return px_begin + tile * kTilePx;
        }
    };
    HlCursor cu_fetch = c_first, cu_store = c_first, cu_taps = c_first, cu_gf = c_first;
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
    // STAGED (W % 32 == 0: every tile full, 16-byte aligned): all global traffic through buffer descriptors - a lane's offsets inside a
    // frame are constants, a tile adds ONE scalar offset (round 5: the 64-bit address arithmetic per load / store was a fifth of the
    // vector instructions of a tile, and this kernel is bound by the instructions it issues)
    auto frame_rsrc = [](const void* p_, uint32_t bytes) {
        const uint64_t a = reinterpret_cast<uint64_t>(p_);
        const uint64_t u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32)) << 32) |
                           (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a);
        return __builtin_amdgcn_make_buffer_rsrc((void*)u, 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rs_cur = PMIN ? frame_rsrc(static_cast<const _Float16*>(cur_) + (size_t)t * HW * kHlK, (uint32_t)HW * 256u)
                                               : frame_rsrc(cur + (size_t)t * 128 * HW, (uint32_t)HW * 512u);
    const __amdgpu_buffer_rsrc_t rs_cur_lo = PMIN ? frame_rsrc(static_cast<const _Float16*>(cur_lo_) + (size_t)t * HW * kHlK, (uint32_t)HW * 256u) : rs_cur;
    const int prow = tid >> 4, pchunk = tid & 15;                 // PMIN: this thread's (pixel row, 16-byte chunk) of the operand tile
    const __amdgpu_buffer_rsrc_t rs_g = frame_rsrc(STAGED ? gprev + (size_t)t * Hp * Wp * kD : cur, STAGED ? (uint32_t)(Hp * Wp) * 1024u : 0u);
    const __amdgpu_buffer_rsrc_t rs_hi = frame_rsrc(PLANES ? (const void*)(out_hi + (size_t)t * HW * kD) : (const void*)cur, PLANES ? (uint32_t)HW * 512u : 0u);
    const __amdgpu_buffer_rsrc_t rs_lo = frame_rsrc(PLANES ? (const void*)(out_lo + (size_t)t * HW * kD) : (const void*)cur, PLANES ? (uint32_t)HW * 512u : 0u);
    const __amdgpu_buffer_rsrc_t rs_f32 = frame_rsrc(F32OUT ? (const void*)(out_f32 + (size_t)t * HW * kD) : (const void*)cur, F32OUT ? (uint32_t)HW * 1024u : 0u);
    const int vo_c0 = (ch * HW + 4 * pg) * 4, vo_c1 = vo_c0 + HW * 4;
    auto fetch = [&](int tile, auto par) {
        constexpr int P = decltype(par)::value;
        if constexpr (PMIN) {
            const int tpx0 = take_px0(cu_fetch, tile);
            if constexpr (STAGED) {
                const int vo = prow * kHlRowBytes + pchunk * 16;
                c0[P] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_cur, vo, tpx0 * kHlRowBytes, 0));
                c1[P] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_cur_lo, vo, tpx0 * kHlRowBytes, 0));
            } else {
                int px = tpx0 + prow;
                px = px < HW ? px : HW - 1;
                const size_t o = ((size_t)t * HW + px) * kHlK + 8 * pchunk;
                c0[P] = *reinterpret_cast<const f32x4*>(static_cast<const _Float16*>(cur_) + o);
                c1[P] = *reinterpret_cast<const f32x4*>(static_cast<const _Float16*>(cur_lo_) + o);
            }
            return;
        }
        if constexpr (STAGED) {
            const int soff = take_px0(cu_fetch, tile) * 4;
            c0[P] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_cur, vo_c0, soff, 0));
            c1[P] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_cur, vo_c1, soff, 0));
            return;
        }
        const int pp = take_px0(cu_fetch, tile) + 4 * pg;                  // c0: channel ch, c1: channel ch + 1, pixels pp .. pp + 3
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
        if constexpr (PMIN) {
            *reinterpret_cast<f32x4*>(ah + hl_a_off(prow, pchunk)) = c0[P];
            *reinterpret_cast<f32x4*>(al + hl_a_off(prow, pchunk)) = c1[P];
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = 4 * pg + j;
            hl_f16x2 vh, vl;                                                 // (channel ch, channel ch + 1) of pixel row `row`
            hl_split2(hl_f32x2{c0[P][j], c1[P][j]}, vh, vl);
            const int o = hl_a_off(row, ch >> 3) + (ch & 7) * 2;             // (ch even: a 4-byte slot)
            *reinterpret_cast<hl_f16x2*>(ah + o) = vh;
            *reinterpret_cast<hl_f16x2*>(al + o) = vl;
        }
    };
    // out tiles -> HBM, all 512 threads: whole 512-byte (fp32: 1-KiB) pixel rows
    auto store_out = [&](int tile) {
        const int tpx0 = take_px0(cu_store, tile);
        if constexpr (STAGED) {
#pragma unroll
            for (int u = 0; u < (PLANES ? 2 : 0); ++u) {
                const int piece = u * 512 + tid;
                const int row = piece >> 5, cpos = piece & 31;
                const u32x4 vh = *reinterpret_cast<const u32x4*>(smem + Lds::o_hi + row * kRowBytes + cpos * 16);
                const u32x4 vl = *reinterpret_cast<const u32x4*>(smem + Lds::o_lo + row * kRowBytes + cpos * 16);
                const int vo = row * kRowBytes + ((cpos ^ swz(row)) * 16);
                __builtin_amdgcn_raw_buffer_store_b128(vh, rs_hi, vo, tpx0 * kRowBytes, 0);
                __builtin_amdgcn_raw_buffer_store_b128(vl, rs_lo, vo, tpx0 * kRowBytes, 0);
            }
            if constexpr (F32OUT) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int piece = u * 512 + tid;
                    const int row = piece >> 6, cpos = piece & 63;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(smem + Lds::o_f32 + row * 1024 + cpos * 16);
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs_f32, row * 1024 + ((cpos ^ (row & 15)) * 16), tpx0 * 1024, 0);
                }
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < (PLANES ? 2 : 0); ++u) {
            const int piece = u * 512 + tid;                            // [row][chunk position]
            const int row = piece >> 5, cpos = piece & 31;
            const int px = tpx0 + row;
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
                const int px = tpx0 + row;
                const u32x4 v = *reinterpret_cast<const u32x4*>(smem + Lds::o_f32 + row * 1024 + cpos * 16);
                if (px < px_end) *reinterpret_cast<u32x4*>(out_f32 + ((size_t)t * HW + px) * kD + ((cpos ^ (row & 15)) * 4)) = v;
            }
        }
    };

    // STAGED: which source rows of g tile number `tile` has to bring - both (the first tile of the workgroup or of a strip: 36 items =
    // [row][18 columns]), the lower one only (18 items) or none (the rows of the tile above are its rows); wave w stages items w, w + 8, ..
    // One 1-KiB row of g per wave instruction.
    struct GNeed { int n, y0, y1, c_lo; };
    auto g_need = [&](HlCursor& cu, int tile) {
        const int strip = cu.strip, y = cu.y;
        cu.next(H);
        const int y0 = (y > 0 ? y - 1 : 0) >> 1, y1 = y0 + 1 < Hp ? y0 + 1 : Hp - 1;       // = floor(max((y + 0.5) / 2 - 0.5, 0)) and the row below
        const int py0 = (y > 1 ? y - 2 : 0) >> 1, py1 = py0 + 1 < Hp ? py0 + 1 : Hp - 1;    // rows of output row y - 1
        GNeed nd;
        nd.n = (tile == 0 || y == 0) ? 2 * Lds::kGCols : (y1 != py1 ? Lds::kGCols : 0);
        nd.y0 = y0;
        nd.y1 = y1;
        nd.c_lo = 16 * strip - 1 > 0 ? 16 * strip - 1 : 0;
        return nd;
    };
    f32x4 gt[2][5];
    GNeed ndv[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};                   // what register set P holds (planned once, at the request)
    auto fetch_g = [&](int tile, auto par) {
        constexpr int P = decltype(par)::value;
        const GNeed nd = g_need(cu_gf, tile);
        ndv[P] = nd;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int item = w + 8 * i;                    // wave-uniform
            if (item < nd.n) {
                const int upper = nd.n > Lds::kGCols ? (item < Lds::kGCols ? 1 : 0) : 0;      // 36 items: the first 18 are row y0
                int col = nd.c_lo + (item < Lds::kGCols ? item : item - Lds::kGCols);
                col = col < Wp ? col : Wp - 1;
                gt[P][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_g, lane * 16, ((upper ? nd.y0 : nd.y1) * Wp + col) * 1024, 0));
            }
        }
    };
    auto commit_g = [&](int tile, auto par) {
        constexpr int P = decltype(par)::value;
        const GNeed nd = ndv[P];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int item = w + 8 * i;
            if (item < nd.n) {
                const int upper = nd.n > Lds::kGCols ? (item < Lds::kGCols ? 1 : 0) : 0;
                const int ci = item < Lds::kGCols ? item : item - Lds::kGCols;
                const int slot = (upper ? nd.y0 : nd.y1) % Lds::kGRows;
                *reinterpret_cast<f32x4*>(smem + Lds::gtile + (slot * Lds::kGCols + ci) * Lds::kGRow + 16 * lane) = gt[P][i];
            }
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
    if constexpr (STAGED) commit_g(0, I0{});
    // STAGED: the two source rows of this lane's taps blended HORIZONTALLY (16 channels each), kept across tiles: the next output row of the
    // strip usually blends the same two source rows with other vertical weights - every other tile reads no taps at all (the tap reads
    // were 128 KiB of LDS traffic per tile, as much as the operand fragments)
    f32x4 hb[2][4];
    int hb_y0 = -1, hb_y1 = -1;
    // One tile. Invariant at the top: LDS holds the operand tile (and the rows of g) of tile `it`; register set P ^ 1 holds tile it + 1
    // (requested one whole iteration ago); set P is free and takes tile it + 2.
    auto body = [&](int it, auto par) {
        constexpr int P = decltype(par)::value;
        using IP = std::integral_constant<int, P>;
        using IQ = std::integral_constant<int, P ^ 1>;
        K4HL_STAMP(0);
        __syncthreads();                                   // operand tile `it` complete; out tiles of it-1 have been read
        K4HL_STAMP(1);
        if (it + 2 < nt) {
            fetch(it + 2, IP{});
            if constexpr (STAGED) fetch_g(it + 2, IP{});
        }
        K4HL_STAMP(2);
        // ---- per-lane taps of g (non-staged form: this lane's accumulator entries - pixel r, channels 32 w + 8 g + 4 h + j - requested
        // before the MFMAs)
        f32x4 tp[4][4];                                    // [tap][g]
        float h1 = 0.f, w1 = 0.f;
        const char* a00 = nullptr;
        const char* a01 = nullptr;
        int grow1 = 0;
        bool fresh = true;
        if constexpr (TAPS && STAGED) {
            const int y = cu_taps.y, x0t = 32 * cu_taps.strip, x = x0t + r;
            cu_taps.next(H);
            const float sy = fmaxf((y + 0.5f) * 0.5f - 0.5f, 0.f), sx = fmaxf((x + 0.5f) * 0.5f - 0.5f, 0.f);
            const int y0 = (int)sy, xs0 = (int)sx;
            const int y1 = y0 + 1 < Hp ? y0 + 1 : Hp - 1;
            const int xs1 = xs0 + 1 < Wp ? xs0 + 1 : Wp - 1;
            h1 = sy - (float)y0;
            w1 = sx - (float)xs0;
            const int c_lo = x0t / 2 - 1 > 0 ? x0t / 2 - 1 : 0;
            const char* g0 = smem + Lds::gtile + (y0 % Lds::kGRows) * Lds::kGCols * Lds::kGRow + (32 * w + 4 * h) * 4;
            a00 = g0 + (xs0 - c_lo) * Lds::kGRow;
            a01 = g0 + (xs1 - c_lo) * Lds::kGRow;          // (the taps themselves are read from LDS group by group behind the MFMAs)
            grow1 = ((y1 % Lds::kGRows) - (y0 % Lds::kGRows)) * Lds::kGCols * Lds::kGRow;        // from the upper to the lower source row
            fresh = it == 0 || y == 0 || y0 != hb_y0 || y1 != hb_y1;
            hb_y0 = y0;
            hb_y1 = y1;
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
        K4HL_STAMP(3);
        // ---- + up(g) in fp32: w00 a + w01 b + w10 c + w11 d with the four tap weights formed once per lane and tile (round 5: four
        // instead of seven vector instructions per element; torch's upsample_bilinear2d groups the same sum as (1-ly) ((1-lx) a + lx b) +
        // ly ((1-lx) c + lx d) - an fp32 rounding of difference, inside the 5e-6 bound against float64 of tests/test_refprec_gpu.py); split; out tiles
        const float h0 = 1.f - h1, w0 = 1.f - w1;
        const float w00 = h0 * w0, w01 = h0 * w1, w10 = h1 * w0, w11 = h1 * w1;
        if constexpr (TAPS && STAGED) {
            if (fresh) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 ta = *reinterpret_cast<const f32x4*>(a00 + 32 * g), tb = *reinterpret_cast<const f32x4*>(a01 + 32 * g);
                    const f32x4 tc = *reinterpret_cast<const f32x4*>(a00 + grow1 + 32 * g), td = *reinterpret_cast<const f32x4*>(a01 + grow1 + 32 * g);
#pragma unroll
                    for (int j = 0; j < 4; j += 2) {
                        const hl_f32x2 u = hl_f32x2{w0, w0} * hl_f32x2{ta[j], ta[j + 1]}, l = hl_f32x2{w0, w0} * hl_f32x2{tc[j], tc[j + 1]};
                        const hl_f32x2 u2 = __builtin_elementwise_fma(hl_f32x2{w1, w1}, hl_f32x2{tb[j], tb[j + 1]}, u);
                        const hl_f32x2 l2 = __builtin_elementwise_fma(hl_f32x2{w1, w1}, hl_f32x2{td[j], td[j + 1]}, l);
                        hb[0][g][j] = u2[0]; hb[0][g][j + 1] = u2[1];
                        hb[1][g][j] = l2[0]; hb[1][g][j + 1] = l2[1];
                    }
                }
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch0 = 32 * w + 8 * g + 4 * h;
            f16x4 oh, ol;
            f32x4 of;
#pragma unroll
            for (int j = 0; j < 4; j += 2) {                // two channels at a time: packed fp32 / conversion instructions
                hl_f32x2 v = {acc[4 * g + j], acc[4 * g + j + 1]};
                if constexpr (TAPS && STAGED) {
                    // torch's grouping: (1 - ly) ((1 - lx) a + lx b) + ly ((1 - lx) c + lx d)
                    v = __builtin_elementwise_fma(hl_f32x2{h0, h0}, hl_f32x2{hb[0][g][j], hb[0][g][j + 1]}, v);
                    v = __builtin_elementwise_fma(hl_f32x2{h1, h1}, hl_f32x2{hb[1][g][j], hb[1][g][j + 1]}, v);
                } else if constexpr (TAPS) {
                    v = __builtin_elementwise_fma(hl_f32x2{w00, w00}, hl_f32x2{tp[0][g][j], tp[0][g][j + 1]}, v);
                    v = __builtin_elementwise_fma(hl_f32x2{w01, w01}, hl_f32x2{tp[1][g][j], tp[1][g][j + 1]}, v);
                    v = __builtin_elementwise_fma(hl_f32x2{w10, w10}, hl_f32x2{tp[2][g][j], tp[2][g][j + 1]}, v);
                    v = __builtin_elementwise_fma(hl_f32x2{w11, w11}, hl_f32x2{tp[3][g][j], tp[3][g][j + 1]}, v);
                }
                if constexpr (PLANES) {
                    hl_f16x2 vh, vl;
                    hl_split2(v, vh, vl);
                    oh[j] = vh[0]; oh[j + 1] = vh[1];
                    ol[j] = vl[0]; ol[j + 1] = vl[1];
                    const hl_f32x2 back = __builtin_convertvector(vh, hl_f32x2) + __builtin_convertvector(vl, hl_f32x2);
                    of[j] = back[0];                       // the fp32 copy holds exactly the planes' value
                    of[j + 1] = back[1];
                } else {
                    of[j] = v[0];
                    of[j + 1] = v[1];
                }
            }
            const int o = r * kRowBytes + (((ch0 >> 3) ^ swz(r)) * 16) + (ch0 & 7) * 2;
            if constexpr (PLANES) {
                *reinterpret_cast<f16x4*>(smem + Lds::o_hi + o) = oh;
                *reinterpret_cast<f16x4*>(smem + Lds::o_lo + o) = ol;
            }
            if constexpr (F32OUT) *reinterpret_cast<f32x4*>(smem + Lds::o_f32 + r * 1024 + (((ch0 >> 2) ^ (r & 15)) * 16)) = of;
        }
        K4HL_STAMP(4);
        __syncthreads();                                   // out tiles complete; every wave is done reading operand tile `it` (and its taps)
        K4HL_STAMP(5);
        if (it + 1 < nt) {
            commit(IQ{});                                  // operand tile it + 1: its loads were issued a whole iteration ago
            if constexpr (STAGED) commit_g(it + 1, IQ{});
        }
        K4HL_STAMP(6);
        store_out(it);
        K4HL_STAMP(7);
    };
    for (int it = 0; it < nt; it += 2) {
        body(it, I0{});
        if (it + 1 < nt) body(it + 1, I1{});
    }
}

template <bool TAPS, bool F32OUT, bool STAGED = false, bool PLANES = true, bool PMIN = false>
__global__ __launch_bounds__(512) void level_fuse_hl_kernel(const void* __restrict__ cur_, const void* __restrict__ cur_lo_,
                                                            const float* __restrict__ gprev, const _Float16* __restrict__ wb_hi,
                                                            const _Float16* __restrict__ wb_lo, const float* __restrict__ bc,
                                                            _Float16* __restrict__ out_hi, _Float16* __restrict__ out_lo,
                                                            float* __restrict__ out_f32, int H, int W, int tiles_per_chunk) {
    level_fuse_hl_body<TAPS, F32OUT, STAGED, PLANES, PMIN>(blockIdx.y, blockIdx.x, cur_, cur_lo_, gprev, wb_hi, wb_lo, bc, out_hi, out_lo, out_f32,
                                                           H, W, tiles_per_chunk);
}

// ALL orders m = 0 .. n - 1 of a level in ONE launch (round 6): level i of 4 needs G^(m) for m = 0 .. 3 - i, every one a
// K = 128 product on the SAME incoming tile. A workgroup still computes one order of one chunk of tiles (the weights of an order fill the
// matrix waves' registers: two orders per workgroup do not fit, and streaming a second order's 64 KiB of weights per 32-pixel tile would move
// more bytes than the re-read of x_i it saves) - but the n workgroups of a chunk are handed to the SAME XCD back to back (workgroup ids go to
// the XCDs round-robin: ids b and b + 8 meet on one XCD), so they walk the same tiles at the same pace and the incoming map crosses
// HBM -> L2 once instead of n times; one launch tail per level instead of n. Order 0 writes the planes, orders >= 1 fp32 only.
struct FuseHlMultiArgs {
    const float* gprev[4];
    const _Float16* wb_hi[4];
    const _Float16* wb_lo[4];
    const float* bc[4];
    float* out_f32[4];
    _Float16* out_hi;
    _Float16* out_lo;
    int n, chunks, pairs;                     // pairs = frames x chunks
};

template <bool TAPS, bool STAGED, bool PMIN>
__global__ __launch_bounds__(512) void level_fuse_hl_multi_kernel(const void* __restrict__ cur_, const void* __restrict__ cur_lo_, FuseHlMultiArgs a,
                                                                  int H, int W, int tiles_per_chunk) {
    // one-dimensional grid over (frame, chunk) pairs q = t * chunks + c and orders m: id = ((q / 8) * n + m) * 8 + q % 8 - the n orders of a pair
    // are 8 ids apart (the same XCD, back to back), consecutive pairs go round the XCDs as the chunks of the per-order launch do
    const int id = blockIdx.x;
    const int x = id & 7, k = id >> 3;
    const int m = k % a.n, q = (k / a.n) * 8 + x;
    if (q >= a.pairs) return;
    const int t = q / a.chunks, c = q - t * a.chunks;
    if (m == 0) {
        level_fuse_hl_body<TAPS, false, STAGED, true, PMIN>(t, c, cur_, cur_lo_, a.gprev[0], a.wb_hi[0], a.wb_lo[0], a.bc[0], a.out_hi, a.out_lo,
                                                            nullptr, H, W, tiles_per_chunk);
    } else {
        level_fuse_hl_body<TAPS, true, STAGED, false, PMIN>(t, c, cur_, cur_lo_, a.gprev[m], a.wb_hi[m], a.wb_lo[m], a.bc[m], nullptr, nullptr,
                                                            a.out_f32[m], H, W, tiles_per_chunk);
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
#ifdef SVPS_K4HL_STAMP
extern "C" int svps_k4hl_debug_read(unsigned long long* stamps) {
    return (int)hipMemcpyFromSymbol(stamps, HIP_SYMBOL(svps::k4hl_stamps), sizeof(unsigned long long) * 2 * 8 * 8);
}
#endif

static int svps_level_fuse_hl_launch(const void* cur, const void* cur_lo, bool pm, const float* gprev, const void* wb_hi, const void* wb_lo,
                                     const float* bc, void* out_hi, void* out_lo, float* out_f32, int T, int H, int W, void* stream_) {
    if (!cur || (pm && !cur_lo) || !wb_hi || !wb_lo || !bc || (!out_hi != !out_lo) || (!out_hi && !out_f32)) return SVPS_ERR_BAD_ARG;
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
        const bool staged = TAPS && (W & 31) == 0 && HW < (1 << 22);   /* a tile = 32 pixels of one output row; 32-bit frame offsets */ \
        auto kern = pm ? (staged ? svps::level_fuse_hl_kernel<TAPS, F32, TAPS, PL, true> : svps::level_fuse_hl_kernel<TAPS, F32, false, PL, true>) \
                       : (staged ? svps::level_fuse_hl_kernel<TAPS, F32, TAPS, PL, false> : svps::level_fuse_hl_kernel<TAPS, F32, false, PL, false>); \
        static SvpsLdsAttr attr[4];                                                                                                \
        if ((e = attr[(pm ? 2 : 0) + (staged ? 1 : 0)].ensure(reinterpret_cast<const void*>(kern), lds)) != hipSuccess) return (int)e; \
        hipLaunchKernelGGL(kern, dim3(chunks, T), dim3(512), lds, stream, cur, cur_lo, gprev, static_cast<const H16*>(wb_hi),      \
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

// svps_level_fuse_hl_multi_fwd (include/slotvps_hip.h): orders 0 .. n - 1 of a level in one launch (level_fuse_hl_multi_kernel)
extern "C" int svps_level_fuse_hl_multi_fwd(const void* cur, const void* cur_lo, int n, const float* const* gprev, const void* const* wb_hi,
                                            const void* const* wb_lo, const float* const* bc, void* out_hi, void* out_lo,
                                            float* const* out_f32, int T, int H, int W, void* stream_) {
    if (!cur || !wb_hi || !wb_lo || !bc || !out_hi || !out_lo || n < 1 || n > 4 || (n > 1 && !out_f32)) return SVPS_ERR_BAD_ARG;
    if (T <= 0 || H <= 0 || W <= 0) return SVPS_ERR_BAD_SHAPE;
    if ((size_t)H * W > svps::kMaxFramePixels) return SVPS_ERR_BAD_SHAPE;
    const bool taps = gprev && gprev[0];
    svps::FuseHlMultiArgs a = {};
    for (int m = 0; m < n; ++m) {
        if (!wb_hi[m] || !wb_lo[m] || !bc[m] || (m > 0 && !out_f32[m]) || (taps != (gprev && gprev[m] != nullptr))) return SVPS_ERR_BAD_ARG;
        a.gprev[m] = taps ? gprev[m] : nullptr;
        a.wb_hi[m] = static_cast<const _Float16*>(wb_hi[m]);
        a.wb_lo[m] = static_cast<const _Float16*>(wb_lo[m]);
        a.bc[m] = bc[m];
        a.out_f32[m] = m > 0 ? out_f32[m] : nullptr;
    }
    if (taps && ((H & 1) || (W & 1))) return SVPS_ERR_BAD_SHAPE;
    a.out_hi = static_cast<_Float16*>(out_hi);
    a.out_lo = static_cast<_Float16*>(out_lo);
    a.n = n;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const bool pm = cur_lo != nullptr;
    const int HW = H * W;
    const int tiles = (HW + svps::kTilePx - 1) / svps::kTilePx;
    int chunks = svps_pick_chunks(T * n, tiles, svps_num_cus());     // n workgroups per chunk share the chip
    const int tpc = (tiles + chunks - 1) / chunks;
    chunks = (tiles + tpc - 1) / tpc;
    a.chunks = chunks;
    a.pairs = T * chunks;
    const int gx = ((a.pairs + 7) / 8) * 8 * n;
    constexpr int lds = svps::FuseHlLds::total;
    const bool staged = taps && (W & 31) == 0 && HW < (1 << 22);
    hipError_t e = hipSuccess;
    svps_prof_mark(SVPS_KERNEL_LEVEL_FUSE, 0, stream);
#define SVPS_LFM(TAPS, STAGED, PM)                                                                                          \
    do {                                                                                                                    \
        auto kern = svps::level_fuse_hl_multi_kernel<TAPS, STAGED, PM>;                                                      \
        static SvpsLdsAttr attr;                                                                                            \
        if ((e = attr.ensure(reinterpret_cast<const void*>(kern), lds)) != hipSuccess) return (int)e;                       \
        hipLaunchKernelGGL(kern, dim3(gx), dim3(512), lds, stream, cur, cur_lo, a, H, W, tpc);                              \
    } while (0)
    if (taps) {
        if (staged) { if (pm) SVPS_LFM(true, true, true); else SVPS_LFM(true, true, false); }
        else { if (pm) SVPS_LFM(true, false, true); else SVPS_LFM(true, false, false); }
    } else {
        if (pm) SVPS_LFM(false, false, true); else SVPS_LFM(false, false, false);
    }
#undef SVPS_LFM
    e = hipGetLastError();
    svps_prof_mark(SVPS_KERNEL_LEVEL_FUSE, 1, stream);
    return (int)e;
}

extern "C" int svps_level_fuse_hl_fwd(const float* cur, const float* gprev, const void* wb_hi, const void* wb_lo, const float* bc,
                                      void* out_hi, void* out_lo, float* out_f32, int T, int H, int W, void* stream_) {
    return svps_level_fuse_hl_launch(cur, nullptr, false, gprev, wb_hi, wb_lo, bc, out_hi, out_lo, out_f32, T, H, W, stream_);
}

// svps_level_fuse_hl_pm_fwd (include/slotvps_hip.h): the same with the incoming map as fp16 hi + lo pixel-major planes [T, H*W, 128]
extern "C" int svps_level_fuse_hl_pm_fwd(const void* cur_hi, const void* cur_lo, const float* gprev, const void* wb_hi, const void* wb_lo,
                                         const float* bc, void* out_hi, void* out_lo, float* out_f32, int T, int H, int W, void* stream_) {
    return svps_level_fuse_hl_launch(cur_hi, cur_lo, true, gprev, wb_hi, wb_lo, bc, out_hi, out_lo, out_f32, T, H, W, stream_);
}
