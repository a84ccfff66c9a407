// K4 - level fusion of the multi-scale slot head for gfx950.
//
// Replaces, for all T frames of a pyramid level in one launch, the feature-side glue of
// MultiScaleDynamicMaskHead.forward (mmdet/models/detectors/dynamic_mask_head.py:171-188):
//
//     level i > 0:  f_i = conv1x1_{384->256}( cat( bilinear_x2(f_{i-1}), x_i ) )        :178-181
//     level 0:      f_0 = conv1x1_{384->256}( cat( x_0, x_0, x_0 ) )                    :182-185
//
// with the SAME (shared) 1x1 conv_trans. F.interpolate(scale 2, bilinear, align_corners=False):
// source coordinate (d + 0.5) / 2 - 0.5 clamped at 0, taps i0 = floor, i1 = min(i0 + 1, n - 1).
// Output: the bf16 pixel-major [T, H*W, 256] map that K3, the next level and K2 consume.
//
// Storage policy: the 384-channel operand (fp32 blend of the four bf16 taps, resp. the incoming
// 128-channel map) and the weight matrix are rounded to bf16 for the matrix cores; accumulation and
// bias are fp32; the result is rounded to bf16 once.
//
// Mapping: 8 waves, wave w owns output channels [32w, 32w+32) - its 32 x 384 weight block stays in 96
// VGPRs as MFMA A fragments. Pixels stream in 32-pixel tiles: all threads build the [32 px][384 ch]
// operand tile in LDS (taps of tile i+1 are fetched into registers while tile i is on the matrix
// cores), 24 MFMA 32x32x16 per wave and tile, results go through an LDS out-tile so HBM sees whole
// 512-byte pixel rows. The incoming 128-channel map is read in the reference's own layout
// ([T, 128, H, W] fp32, NCHW) or as bf16 pixel-major [T, H*W, 128].
//
// Roofline: HBM - per output pixel 512 B (fp32 NCHW input) or 256 B (bf16) in, 512 B out; the four
// upsampling taps come from the 4x smaller previous level (L2 / Infinity Cache resident).
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "../../include/slotvps_hip.h"

namespace svps {

constexpr int kFuseIn = 384;                  // concatenated input channels
constexpr int kFuseRowBytes = kFuseIn * 2;    // 768 B per pixel row of the operand tile

struct FuseLds {
    static constexpr int atile = 0;                                 // [32][384] bf16, 16-B chunks swizzled
    static constexpr int otile = kTilePx * kFuseRowBytes;           // [32][256] bf16 out tile
    static constexpr int total = otile + kTileBytes;
};

// chunk swizzle of the 48-chunk operand rows: XOR on the low four bits keeps chunks inside their
// 16-chunk group, conflict-free for ds_read_b128 over 16 rows
// fp32 accumulator -> map element. fp16 maps SATURATE at +-65 504 (an overflow to inf would turn into NaN in every consumer; the bf16
// form's consumers saturate when they convert the map to fp16 in LDS - same behaviour either way)
// BP ("bf16 precision", fp16 maps only): the value is rounded to bf16 FIRST and then stored in the fp16 encoding - the storage policy
// of the bf16 form (the same values, bit for bit, above fp16's subnormal range) in the encoding every consumer's matrix instructions
// take directly, so that their bf16 -> fp16 pass over the tile in LDS disappears (MultiScaleDynamicMaskHead.map_dtype = "bf16").
template <typename MT, bool BP = false>
__device__ __forceinline__ MT to_map(float x) {
    if constexpr (__is_same(MT, _Float16)) {
        if constexpr (BP) x = (float)(__bf16)x;
        x = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
    }
    return (MT)x;
}
// two at a time. BP form: ONE v_cvt_pk_bf16_f32 for the pair, two unpack instructions, one v_cvt_pkrtz_f16_f32 - round-to-zero is exact on
// bf16 values inside fp16's normal range and saturates at 65 504 by itself (four instructions per pair instead of seven)
template <typename MT, bool BP = false>
__device__ __forceinline__ void to_map2(float a, float b, MT& ra, MT& rb) {
    if constexpr (__is_same(MT, _Float16) && BP) {
        typedef __fp16 hf2_t __attribute__((ext_vector_type(2)));
        uint32_t w;                                              // (hipcc converts the two halves with two instructions if left to itself)
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(a), "v"(b));
        const hf2_t h2 = __builtin_amdgcn_cvt_pkrtz(__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u));
        const uint32_t hw = __builtin_bit_cast(uint32_t, h2);
        ra = __builtin_bit_cast(_Float16, (uint16_t)(hw & 0xffffu));
        rb = __builtin_bit_cast(_Float16, (uint16_t)(hw >> 16));
    } else {
        ra = to_map<MT, BP>(a);
        rb = to_map<MT, BP>(b);
    }
}

__device__ __forceinline__ int a_off(int row, int chunk) {
    return row * kFuseRowBytes + (((chunk & ~15) | ((chunk ^ swz(row)) & 15)) * 16);
}

template <typename MT, bool NCHW_F32, bool LEVEL0, bool BP = false>
__global__ __launch_bounds__(512) void level_fuse_kernel(
    const void* __restrict__ cur_,        // [T, 128, H, W] fp32 (NCHW_F32) or [T, H*W, 128] bf16
    const MT* __restrict__ prev,      // [T, (H/2)*(W/2), 256] bf16 pixel-major (unused for LEVEL0)
    const MT* __restrict__ wc,        // [256, 384] conv weight (row = output channel), element type MT - BP: bf16 (converted on load)
    const float* __restrict__ bc,         // [256]
    MT* __restrict__ out,             // [T, H*W, 256]
    int H, int W, int tiles_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef MT mx8 __attribute__((ext_vector_type(8)));         // MT: element type of the maps and of the conv operands (common.h)
    typedef MT mx4 __attribute__((ext_vector_type(4)));
    // a pixel-major 16-bit incoming map has the element type of the conv's operands in level_fuse_kernel_v4: MT, but bf16 in the BP
    // form (here re-encoded on the way into the operand tile, like the weights)
    using Lds = FuseLds;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r_ = lane & 31, h_ = lane >> 5;
    const int t = blockIdx.y, c = blockIdx.x;
    const int HW = H * W;
    const int Hp = H >> 1, Wp = W >> 1;

    const int px_begin = c * tiles_per_chunk * kTilePx;
    int px_end = px_begin + tiles_per_chunk * kTilePx;
    px_end = px_end < HW ? px_end : HW;
    const int nt = (px_end - px_begin + kTilePx - 1) / kTilePx;

    // ---- weight block of this wave: rows 32w .. 32w+31, 24 k-steps ----------------------------------
    mx8 wf[24];
    {
        const MT* row = wc + (size_t)(32 * w + r_) * kFuseIn + 8 * h_;
#pragma unroll
        for (int ks = 0; ks < 24; ++ks) {
            if constexpr (BP) {      // the weights of the bf16-in-fp16 form are bf16 (level_fuse_kernel_v4 uses them as they are): re-encoded once
                const bf16x8 wb = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(row + 16 * ks));
#pragma unroll
                for (int j = 0; j < 8; ++j) wf[ks][j] = (MT)(float)wb[j];
            } else {
                wf[ks] = __builtin_bit_cast(mx8, *reinterpret_cast<const u32x4*>(row + 16 * ks));
            }
        }
    }
    float bias[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) bias[i] = bc[32 * w + acc_row(i, h_)];

    // ---- operand-tile builders -------------------------------------------------------------------
    // (a) incoming 128-channel map -> chunks 32..47 (channels 256..383); LEVEL0 also -> 0..15, 16..31
    //     NCHW fp32: thread = (channel, 8-pixel group): 2 x 16-B loads, 8 two-byte LDS stores
    //     pixel-major bf16: thread = (pixel, 16-B chunk): one 16-B load, one 16-B LDS store
    // (b) upsampled previous level -> chunks 0..31: thread = (pixel, 2 chunks), 4 taps each
    struct Pre {
        f32x4 c0, c1;           // NCHW path
        u32x4 cb;               // bf16 path
        u32x4 tap[2][4];        // [chunk][tap]
        float wy, wx;           // lambda_y, lambda_x of this thread's pixel
    };
    auto prefetch = [&](int tile, Pre& p) {
        const int px0 = px_begin + tile * kTilePx;
        if constexpr (NCHW_F32) {
            const int ch = tid >> 2, pg = tid & 3;                     // channel 0..127, pixel group 0..3
            int pp = px0 + 8 * pg;
            const float* src = static_cast<const float*>(cur_) + ((size_t)t * 128 + ch) * HW;
            if (pp + 8 <= HW) {
                p.c0 = *reinterpret_cast<const f32x4*>(src + pp);
                p.c1 = *reinterpret_cast<const f32x4*>(src + pp + 4);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    p.c0[j] = src[pp + j < HW ? pp + j : HW - 1];
                    p.c1[j] = src[pp + 4 + j < HW ? pp + 4 + j : HW - 1];
                }
            }
        } else {
            const int px = tid >> 4, ck = tid & 15;
            int pp = px0 + px;
            pp = pp < HW ? pp : HW - 1;
            p.cb = *reinterpret_cast<const u32x4*>(static_cast<const MT*>(cur_) + ((size_t)t * HW + pp) * 128 + 8 * ck);
            if constexpr (BP) {
                const bf16x8 xb = __builtin_bit_cast(bf16x8, p.cb);
                mx8 xm;
#pragma unroll
                for (int j = 0; j < 8; ++j) xm[j] = to_map<MT, false>((float)xb[j]);
                p.cb = __builtin_bit_cast(u32x4, xm);
            }
        }
        if constexpr (!LEVEL0) {
            const int px = tid >> 4, ck = tid & 15;                    // chunks ck and ck + 16
            int pp = px0 + px;
            pp = pp < HW ? pp : HW - 1;
            const int y = pp / W, x = pp - y * W;
            const float sy = fmaxf((y + 0.5f) * 0.5f - 0.5f, 0.f), sx = fmaxf((x + 0.5f) * 0.5f - 0.5f, 0.f);
            const int y0 = (int)sy, x0 = (int)sx;
            const int y1 = y0 + 1 < Hp ? y0 + 1 : Hp - 1, x1 = x0 + 1 < Wp ? x0 + 1 : Wp - 1;
            p.wy = sy - (float)y0;
            p.wx = sx - (float)x0;
            const MT* pb = prev + (size_t)t * Hp * Wp * kD;
            const MT* t00 = pb + ((size_t)y0 * Wp + x0) * kD;
            const MT* t01 = pb + ((size_t)y0 * Wp + x1) * kD;
            const MT* t10 = pb + ((size_t)y1 * Wp + x0) * kD;
            const MT* t11 = pb + ((size_t)y1 * Wp + x1) * kD;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int co = 8 * (ck + 16 * u);
                p.tap[u][0] = *reinterpret_cast<const u32x4*>(t00 + co);
                p.tap[u][1] = *reinterpret_cast<const u32x4*>(t01 + co);
                p.tap[u][2] = *reinterpret_cast<const u32x4*>(t10 + co);
                p.tap[u][3] = *reinterpret_cast<const u32x4*>(t11 + co);
            }
        }
    };
    auto commit = [&](const Pre& p) {
        char* at = smem + Lds::atile;
        if constexpr (NCHW_F32) {
            const int ch = tid >> 2, pg = tid & 3;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = 8 * pg + j;
                const MT val = to_map<MT, BP>(j < 4 ? p.c0[j] : p.c1[j - 4]);
                const int chunk = 32 + (ch >> 3);
                *reinterpret_cast<MT*>(at + a_off(row, chunk) + (ch & 7) * 2) = val;
                if constexpr (LEVEL0) {
                    *reinterpret_cast<MT*>(at + a_off(row, chunk - 32) + (ch & 7) * 2) = val;
                    *reinterpret_cast<MT*>(at + a_off(row, chunk - 16) + (ch & 7) * 2) = val;
                }
            }
        } else {
            const int px = tid >> 4, ck = tid & 15;
            *reinterpret_cast<u32x4*>(at + a_off(px, 32 + ck)) = p.cb;
            if constexpr (LEVEL0) {
                *reinterpret_cast<u32x4*>(at + a_off(px, ck)) = p.cb;
                *reinterpret_cast<u32x4*>(at + a_off(px, 16 + ck)) = p.cb;
            }
        }
        if constexpr (!LEVEL0) {
            const int px = tid >> 4, ck = tid & 15;
            // torch's upsample_bilinear2d: (1-ly) * ((1-lx) a + lx b) + ly * ((1-lx) c + lx d) in fp32
            const float h1 = p.wy, h0 = 1.f - p.wy, w1 = p.wx, w0 = 1.f - p.wx;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const mx8 a = __builtin_bit_cast(mx8, p.tap[u][0]), b = __builtin_bit_cast(mx8, p.tap[u][1]);
                const mx8 cc = __builtin_bit_cast(mx8, p.tap[u][2]), d = __builtin_bit_cast(mx8, p.tap[u][3]);
                mx8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    o[j] = to_map<MT, BP>(h0 * (w0 * (float)a[j] + w1 * (float)b[j]) + h1 * (w0 * (float)cc[j] + w1 * (float)d[j]));
                *reinterpret_cast<mx8*>(at + a_off(px, ck + 16 * u)) = o;
            }
        }
    };
    // out tile -> HBM, all 512 threads: 16 KiB per tile, 2 x 16 B per thread
    auto store_out = [&](int tile) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int piece = u * 512 + tid;                            // [row][chunk position]
            const int row = piece >> 5, cpos = piece & 31;
            const int px = px_begin + tile * kTilePx + row;
            const u32x4 val = *reinterpret_cast<const u32x4*>(smem + Lds::otile + row * kRowBytes + cpos * 16);
            if (px < px_end)
                *reinterpret_cast<u32x4*>(out + ((size_t)t * HW + px) * kD + ((cpos ^ swz(row)) * 8)) = val;
        }
    };

    Pre pre;
    prefetch(0, pre);
    commit(pre);
    for (int it = 0; it < nt; ++it) {
        __syncthreads();                                   // a(it): operand tile it built, out tile it-1 complete
        // loads first: vmcnt retires in order, so the wait for the taps must not sit behind the stores of the previous tile
        if (it + 1 < nt) prefetch(it + 1, pre);            // taps of the next tile fly under the MFMAs
        if (it >= 1) store_out(it - 1);
        int r = r_, h = h_;
        asm volatile("" : "+v"(r), "+v"(h));
        const char* at = smem + Lds::atile;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = bias[i];
#pragma unroll
        for (int grp = 0; grp < 3; ++grp) {
            mx8 xf[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) xf[u] = *reinterpret_cast<const mx8*>(at + a_off(r, 2 * (8 * grp + u) + h));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = mfma16(wf[8 * grp + u], xf[u], acc);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                                   // b(it): every wave is done reading operand tile it
        char* ot = smem + Lds::otile;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            mx4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = to_map<MT, BP>(acc[4 * g + j]);
            const int ch0 = 32 * w + 8 * g + 4 * h;
            *reinterpret_cast<mx4*>(ot + r * kRowBytes + (((ch0 >> 3) ^ swz(r)) * 16) + (ch0 & 7) * 2) = o;
        }
        if (it + 1 < nt) commit(pre);
    }
    __syncthreads();
    store_out(nt - 1);
}


// ---------------------------------------------------------------------------------------------------------------
// Fast path for levels > 0 with W % 32 == 0 (every headline level): a tile is 32 consecutive pixels of ONE output row,
// so its bilinear taps come from two rows x 18 columns of the previous level. Those 18 KiB are staged once per tile by
// LDS-DMA (inline asm, 18 instructions per workgroup) and blended from LDS, instead of 8 scattered 16-byte global loads
// per thread (64 KiB per tile through the texture-address path, 3.5x redundant). The incoming NCHW fp32 map is read
// by (channel pair, pixel quad) threads - two 16-byte loads, four packed 4-byte LDS writes - and the out tile leaves
// through buffer stores with linear addressing (hardware range check, no per-store address arithmetic).
// Order of vector-memory operations per iteration: DMA (asm) and the map loads of tile it+1 first, the stores of tile
// it-1 last: vmcnt retires in order, so `vmcnt(#stores)` before barrier b covers every load without draining the stores.
struct Fuse2Lds {
    // operand and out tiles with PADDED rows (784 B / 528 B) instead of the chunk swizzle: every LDS address is then
    // "lane base + compile-time constant"; the 16-byte fragment reads of 16 consecutive rows still hit 16 distinct bank groups
    static constexpr int kARow = kFuseRowBytes + 16;
    static constexpr int kORow = kRowBytes + 16;
    static constexpr int atile = 0;                                 // [32][384] bf16 operand tile
    static constexpr int otile = kTilePx * kARow;                   // [32][256] bf16 out tile
    static constexpr int stage = otile + kTilePx * kORow;           // [2 rows][18 px][512 B] taps of the next tile
    static constexpr int kStageCols = 18;
    static constexpr int stage_bytes = 2 * kStageCols * kRowBytes;
    static constexpr int cur = stage + 2 * stage_bytes;              // incoming map of a tile as it lies in memory: [128][32] fp32 or [32][128] bf16
    static constexpr int cur_bytes = 128 * kTilePx * 4;
    static constexpr int total = cur + 2 * cur_bytes;                // taps and map are requested two tiles ahead: double-buffered
};

template <bool NCHW_F32, int ABL = 0>
__global__ __launch_bounds__(512) void level_fuse_kernel_v2(
    const void* __restrict__ cur_, const __bf16* __restrict__ prev, const __bf16* __restrict__ wc,
    const float* __restrict__ bc, __bf16* __restrict__ out, int H, int W, int tiles_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using Lds = Fuse2Lds;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r_ = lane & 31, h_ = lane >> 5;
    const int t = blockIdx.y, c = blockIdx.x;
    const int HW = H * W;
    const int Hp = H >> 1, Wp = W >> 1;
    const int tiles = HW / kTilePx;                                  // W % 32 == 0: no ragged tile
    const int tile_begin = c * tiles_per_chunk;
    int tile_end = tile_begin + tiles_per_chunk;
    tile_end = tile_end < tiles ? tile_end : tiles;
    const int nt = tile_end - tile_begin;
    // Tile order = COLUMN STRIPS: the g-th tile of the frame is strip g / H (32 output columns), row g % H, so a workgroup
    // walks DOWN a strip. Consecutive tiles then share their tap rows (output rows 2m+1 and 2m+2 read the same two source
    // rows, 2m+3 one of them): with row-major order the second use came a whole image row of streaming later
    // (32 workgroups x 0.5 MB against a 4 MB L2) and was fetched again - 1.33x the algorithmic bytes left L2; now it comes
    // one tile later. The out tile is still one contiguous 16 KiB block, the incoming map is read as before.
    const int tiles_per_row = W / kTilePx;
    (void)tiles_per_row;
    auto tile_px0 = [&](int tile) {
        const int g = tile_begin + tile;
        const int strip = g / H;
        return (g - strip * H) * W + strip * kTilePx;
    };

    bf16x8 wf[24];
    {
        const __bf16* row = wc + (size_t)(32 * w + r_) * kFuseIn + 8 * h_;
#pragma unroll
        for (int ks = 0; ks < 24; ++ks)
            wf[ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(row + 16 * ks));
    }
    float bias[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) bias[i] = bc[32 * w + acc_row(i, h_)];

    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>((SVPS_LDS const void*)smem);
    u32x4 psrd;
    {
        const uint64_t a = reinterpret_cast<uint64_t>(prev + (size_t)t * Hp * Wp * kD);
        psrd[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
        psrd[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
        psrd[2] = __builtin_amdgcn_readfirstlane((uint32_t)(Hp * Wp) * kRowBytes);
        psrd[3] = 0x00020000u;
    }
    const __amdgpu_buffer_rsrc_t osrd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(out + (size_t)t * HW * kD), 0, HW * kRowBytes, 0x00020000);

    // horizontal blend weights as MFMA B fragments (tile-invariant: a tile starts at an even column, the staged columns
    // start one source pixel to its left and are clamped into the row when they are read from memory): pixel n takes
    // staged columns c0 = (n + 1) >> 1 and c0 + 1 with weights (1 - lx, lx), lx = 0.25 for odd n, 0.75 for even n.
    // Lane (n, h) holds taps 16 ks + 8 h + j, j = 0..7.
    bf16x8 bwx[2];
    {
        const int c0 = (r_ + 1) >> 1;
        const float lx = (r_ & 1) ? 0.25f : 0.75f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kk = 16 * ks + 8 * h_ + j;
                bwx[ks][j] = (__bf16)(kk == c0 ? 1.f - lx : (kk == c0 + 1 ? lx : 0.f));
            }
    }
    // A fragments = staged taps transposed, channels 32w .. 32w+31: byte offsets of the two transposed reads of k-step ks
    // inside one staged source row (see read_col_frag in common.h); taps past column 17 carry weight 0 and are clamped
    // onto column 17 so that they read finite data.
    int tap_off[2][2];
    {
        const int gq = lane >> 4, i = lane & 15, q = i >> 2, p4 = i & 3;
        const int chunk = 4 * w + 2 * (gq & 1) + (p4 >> 1), sub = 8 * (p4 & 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                int row = 16 * ks + 8 * (gq >> 1) + q + 4 * half;
                row = row < Fuse2Lds::kStageCols ? row : Fuse2Lds::kStageCols - 1;
                tap_off[ks][half] = row * kRowBytes + ((chunk ^ swz(row)) * 16) + sub;
            }
    }

    // tile geometry (wave-uniform): output row y, first column x0; source rows ys0 / ys1 with weight wy of ys1;
    // staged columns xs_base .. xs_base + 17 (clamped into the row when read from memory)
    struct Geo { int ys0, ys1, xs_base, x0; float wy; };
    auto geometry = [&](int tile) {
        Geo g;
        const int px0 = tile_px0(tile);
        const int y = px0 / W;
        g.x0 = px0 - y * W;
        const float sy = fmaxf((y + 0.5f) * 0.5f - 0.5f, 0.f);
        g.ys0 = (int)sy;
        g.ys1 = g.ys0 + 1 < Hp ? g.ys0 + 1 : Hp - 1;
        g.wy = sy - (float)g.ys0;
        g.xs_base = (g.x0 >> 1) - 1;
        return g;
    };
    // DMA instruction q (0..17) of a tile: source row q / 9, staged columns 2 (q % 9) and 2 (q % 9) + 1; wave w issues q = w, w + 8, w + 16
    auto stage_taps = [&](int tile) {
        if constexpr (ABL & 8) return;
        const Geo g = geometry(tile);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int q = w + 8 * k;
            if (q >= 18) break;
            const int row = q / 9, cp = q - 9 * row;
            int col = g.xs_base + 2 * cp + h_;
            col = col < 0 ? 0 : (col < Wp ? col : Wp - 1);
            const int ys = row ? g.ys1 : g.ys0;
            // 16-byte chunks XOR-swizzled by the staged column index (on the source side, as in K1): the transposed
            // fragment reads of the blend then touch distinct banks
            const int voff = (ys * Wp + col) * kRowBytes + (((lane & 31) ^ swz(2 * cp + h_)) * 16);
            const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + Lds::stage + (tile & 1) * Lds::stage_bytes +
                                                                (row * Lds::kStageCols + 2 * cp) * kRowBytes);
            uint32_t keep;
            asm volatile(
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b32 m0, %1\n\t"
                "s_nop 0\n\t"
                "buffer_load_dwordx4 %2, %3, 0 offen lds\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(keep)
                : "s"(dst), "v"(voff), "s"(psrd)
                : "memory");
        }
    };

    // incoming map of a tile -> LDS by DMA, in its memory layout. NCHW fp32: one instruction = 8 channels x 32 pixels
    // (lane = (channel, pixel quad)), 16 per tile, wave w issues channels 8w.. and 64 + 8w..; pixel-major bf16: the tile is
    // one contiguous 8 KiB block, one instruction per wave.
    u32x4 csrd;
    {
        const size_t frame = NCHW_F32 ? (size_t)128 * HW * 4 : (size_t)HW * 256;
        const uint64_t a = reinterpret_cast<uint64_t>(static_cast<const char*>(cur_) + (size_t)t * frame);
        csrd[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
        csrd[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
        csrd[2] = __builtin_amdgcn_readfirstlane((uint32_t)frame);
        csrd[3] = 0x00020000u;
    }
    constexpr int kCurDma = NCHW_F32 ? 2 : 1;            // map DMA instructions per wave and tile
    auto stage_cur = [&](int tile) {
        if constexpr (ABL & 8) return;
        const int px0 = tile_px0(tile);
        const uint32_t base = lds0 + Lds::cur + (tile & 1) * Lds::cur_bytes;
#pragma unroll
        for (int k = 0; k < kCurDma; ++k) {
            int voff, soff;
            uint32_t dst;
            if constexpr (NCHW_F32) {
                const int ch = 8 * w + 64 * k + (lane >> 3);
                voff = (ch * HW + 4 * (lane & 7)) * 4;
                soff = __builtin_amdgcn_readfirstlane(px0 * 4);
                dst = __builtin_amdgcn_readfirstlane(base + (8 * w + 64 * k) * kTilePx * 4);
            } else {
                voff = w * 1024 + lane * 16;
                soff = __builtin_amdgcn_readfirstlane(px0 * 256);
                dst = __builtin_amdgcn_readfirstlane(base + w * 1024);
            }
            uint32_t keep;
            asm volatile(
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b32 m0, %1\n\t"
                "s_nop 0\n\t"
                "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(keep)
                : "s"(dst), "v"(voff), "s"(csrd), "s"(soff)
                : "memory");
        }
    };
    auto commit = [&](int tile) {
        char* at = smem + Lds::atile;
        const char* cs = smem + Lds::cur + (tile & 1) * Lds::cur_bytes;
        struct { f32x4 c0, c1; u32x4 cb; } p;
        if constexpr (NCHW_F32) {
            const int cp = tid >> 3, pq = tid & 7;                     // channels 2cp, 2cp + 1; pixels 4pq .. 4pq + 3
            p.c0 = *reinterpret_cast<const f32x4*>(cs + (2 * cp) * kTilePx * 4 + pq * 16);
            p.c1 = *reinterpret_cast<const f32x4*>(cs + (2 * cp + 1) * kTilePx * 4 + pq * 16);
            const int chunk = 32 + (cp >> 2), sub = (cp & 3) * 4;      // 16-byte chunk of channels 256 + 2cp, byte inside it
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
                bf16x2 v2;
                v2[0] = (__bf16)p.c0[j];
                v2[1] = (__bf16)p.c1[j];
                *reinterpret_cast<bf16x2*>(at + (4 * pq + j) * Lds::kARow + chunk * 16 + sub) = v2;
            }
        } else {
            const int px = tid >> 4, ck = tid & 15;
            p.cb = *reinterpret_cast<const u32x4*>(cs + px * 256 + ck * 16);
            *reinterpret_cast<u32x4*>(at + px * Lds::kARow + (32 + ck) * 16) = p.cb;
        }
        // bilinear x2 on the matrix cores: up[c][px] = h0 * (Wx . row0)[c][px] + h1 * (Wx . row1)[c][px], where Wx[tap][px]
        // holds the two horizontal weights of pixel px (bwx, the same for every tile). Wave w blends channels 32w .. 32w+31
        // of all 32 pixels: A = staged taps transposed (hardware-transposed LDS reads), 2 k-steps of 16 taps per source row.
        // The products (bf16 tap x {0, .25, .75, 1}) are exact and at most two are non-zero per sum, so each row sum is
        // round(w0 a + w1 b) - torch's upsample_bilinear2d expression  (1-ly) ((1-lx) a + lx b) + ly ((1-lx) c + lx d).
        if constexpr (!(ABL & 4)) {
            const Geo g = geometry(tile);
            const float h1 = g.wy, h0 = 1.f - g.wy;
            const int so = Lds::stage + (tile & 1) * Lds::stage_bytes;
            f32x16 up[2];
#pragma unroll
            for (int row = 0; row < 2; ++row) {
                bf16x8 af[2];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const char* base = smem + so + row * Lds::kStageCols * kRowBytes;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((SVPS_LDS bf16x4*)(base + tap_off[ks][0]));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((SVPS_LDS bf16x4*)(base + tap_off[ks][1]));
                    af[ks] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
                f32x16 z;
#pragma unroll
                for (int i = 0; i < 16; ++i) z[i] = 0.f;
                z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bwx[0], z, 0, 0, 0);
                up[row] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bwx[1], z, 0, 0, 0);
            }
            const f32x2 h0v = {h0, h0}, h1v = {h1, h1};
            const int wo = Lds::atile + r_ * Lds::kARow + (32 * w + 4 * h_) * 2;      // pixel row r_, channels 32w + 8g + 4h ..
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const f32x2 top = {up[0][4 * gq + j], up[0][4 * gq + j + 1]}, bot = {up[1][4 * gq + j], up[1][4 * gq + j + 1]};
                    const f32x2 y = __builtin_elementwise_fma(h1v, bot, h0v * top);
                    o[j] = (__bf16)y[0];
                    o[j + 1] = (__bf16)y[1];
                }
                *reinterpret_cast<bf16x4*>(smem + wo + 16 * gq) = o;
            }
        }
    };
    auto store_out = [&](int tile) {                                  // 16 KiB per tile, 2 x 16 B per thread, linear in HBM
        const int base = tile_px0(tile) * kRowBytes + tid * 16;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = 16 * u + (tid >> 5), gc = tid & 31;
            const u32x4 val = *reinterpret_cast<const u32x4*>(smem + Lds::otile + row * Lds::kORow + gc * 16);
            if constexpr (!(ABL & 1)) __builtin_amdgcn_raw_buffer_store_b128(val, osrd, base + u * 8192, 0, 0);
        }
    };

    // Two tiles ahead: taps and map of tile it+2 are requested at the top of iteration it and consumed by commit(it+2)
    // at the end of iteration it+1, so a full tile of work covers their latency. All requests are asm LDS-DMA (invisible
    // to hipcc, which would otherwise wait for them early); the only compiler-visible vector-memory operations are the
    // two buffer stores per thread and tile, which hipcc never waits for.
    const int n_req = (w < 2 ? 3 : 2) + kCurDma;         // DMA instructions of this wave per tile (18 tap pieces over 8 waves)
    stage_taps(0);
    stage_cur(0);
    if (nt > 1) {
        stage_taps(1);
        stage_cur(1);
        wait_vm_dyn(n_req);                              // tile 0 landed; tile 1 may still fly
    } else {
        wait_vm<0>();
    }
    __syncthreads();
    commit(0);
    for (int it = 0; it < nt; ++it) {
        __syncthreads();                                   // a(it): operand tile it built, out tile it-1 complete, requests of tile it consumed
        if (it + 2 < nt) {
            stage_taps(it + 2);
            stage_cur(it + 2);
        }
        int r = r_, h = h_;
        asm volatile("" : "+v"(r), "+v"(h));
        const char* at = smem + Lds::atile;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = bias[i];
#pragma unroll
        for (int grp = 0; grp < ((ABL & 2) ? 0 : 3); ++grp) {
            bf16x8 xf[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) xf[u] = *reinterpret_cast<const bf16x8*>(at + r * Lds::kARow + (2 * (8 * grp + u) + h) * 16);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[8 * grp + u], xf[u], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (it >= 1) store_out(it - 1);                    // after the MFMAs: the address path has drained the requests above
        // tile it+1 landed. Younger, in issue order: stores(it-2) [issued in iteration it-1 after the requests of tile
        // it+1], requests of tile it+2, stores(it-1).
        if (it + 1 < nt) {
            constexpr int kSt = (ABL & 1) ? 0 : 2;
            wait_vm_dyn((it >= 2 ? kSt : 0) + (it + 2 < nt ? n_req : 0) + (it >= 1 ? kSt : 0));
        }
        __syncthreads();                                   // b(it): every wave is done reading operand tile it; tile it+1's requests visible
        char* ot = smem + Lds::otile;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (__bf16)acc[4 * g + j];
            const int ch0 = 32 * w + 8 * g + 4 * h;
            *reinterpret_cast<bf16x4*>(ot + r * Lds::kORow + (ch0 >> 3) * 16 + (ch0 & 7) * 2) = o;
        }
        if (it + 1 < nt) commit(it + 1);
    }
    __syncthreads();
    store_out(nt - 1);
}


// ---------------------------------------------------------------------------------------------------------------
// K4 v4 (round 3): wave-specialised form of the fast path. The SQ counters and an experiment with four extra waves that took over
// every vector-memory instruction of v2 (-5 %) point at LDS traffic: With 8 waves x 32 output channels every wave
// reads the whole 32 x 384 operand tile - 196 KiB of LDS reads per tile for 24 KiB of data, 1 KiB per MFMA, which is exactly
// the LDS peak (128 B / clk) at the full matrix rate - on top of the DMA landing, the operand build and the out tile.
//   waves 0 - 3 ("matrix"): 64 output channels each (weights in 192 registers), every operand fragment feeds TWO MFMAs:
//       48 MFMAs per tile and wave, operand reads halved; nothing else: read A(it), MFMA, write O(it)
//   waves 4 - 7 ("helpers"): everything else - LDS-DMA requests of tile it+2 (taps + incoming map), stores of out tile it-1,
//       operand tile of tile it+1 (conversion of the incoming map, bilinear blend of the taps on the matrix cores)
// one workgroup barrier per tile; operand and out tiles double-buffered (154 KiB of LDS).
#ifdef SVPS_K4_STAMP
// diagnostic build only (tools/k4_stamps.py): s_memtime stamps of one workgroup's matrix wave 0 and helper wave 0, iterations 8 .. 15
__device__ unsigned long long k4_stamps[2][8][8];            // [matrix / helper][iteration - 8][point]
#define K4_STAMP(role, pt)                                                                               \
    do {                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        if (blockIdx.x == 3 && blockIdx.y == 2 && it >= 8 && it < 16 && (threadIdx.x & 255) == 0)        \
            k4_stamps[role][it - 8][pt] = __builtin_amdgcn_s_memtime();                                  \
        __builtin_amdgcn_sched_barrier(0);                                                               \
    } while (0)
#else
#define K4_STAMP(role, pt) do {} while (0)
#endif

struct Fuse4Lds {
    static constexpr int kARow = kFuseRowBytes + 16;
    static constexpr int kORow = kRowBytes + 16;
    static constexpr int a_bytes = kTilePx * kARow;                 // 25 088
    static constexpr int kOWRow = 128 + 16;                         // one pixel's 64 channels of a matrix wave, padded
    static constexpr int o_wave = kTilePx * kOWRow;                 // 4 608: wave-private out block
    static constexpr int atile = 0;                                 // [2][32][384] bf16 operand tiles
    static constexpr int otile = 2 * a_bytes;                       // [4 matrix waves][32 px][64 channels] bf16
    static constexpr int stage = otile + 4 * o_wave;                // [2][2 rows][18 px][512 B] taps
    static constexpr int kStageCols = 18;
    static constexpr int stage_bytes = 2 * kStageCols * kRowBytes;
    static constexpr int bias = stage + 2 * stage_bytes;            // [256] float
    static constexpr int total = bias + 1024;
};
static_assert(Fuse4Lds::total <= 160 * 1024, "LDS layout");

template <typename MT, bool NCHW_F32, bool BP = false>
__global__ __launch_bounds__(512) void level_fuse_kernel_v4(
    const void* __restrict__ cur_, const MT* __restrict__ prev, const void* __restrict__ wc_,
    const float* __restrict__ bc, MT* __restrict__ out, int H, int W, int tiles_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef MT mx8 __attribute__((ext_vector_type(8)));         // MT: element type of the maps (taps of the previous level, result)
    typedef MT mx4 __attribute__((ext_vector_type(4)));
    // OT: element type of the conv's operands (operand tile in LDS, weights). BP (bf16 values in the fp16 encoding): the operands ARE
    // bf16 - one rounding, the conv on bf16 MFMAs as in the bf16 form - and only the result is re-encoded (matrix waves, which have slack)
    using OT = typename std::conditional<BP, __bf16, MT>::type;
    typedef OT ox8 __attribute__((ext_vector_type(8)));
    typedef OT ox4 __attribute__((ext_vector_type(4)));
    const OT* wc = static_cast<const OT*>(wc_);
    // (a pixel-major 16-bit incoming map is copied into the operand tile as it is: its element type is OT)
    using Lds = Fuse4Lds;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r_ = lane & 31, h_ = lane >> 5;
    const int t = blockIdx.y, c = blockIdx.x;
    const int HW = H * W;
    const int Hp = H >> 1, Wp = W >> 1;
    const int tiles = HW / kTilePx;                                  // W % 32 == 0: no ragged tile
    const int tile_begin = c * tiles_per_chunk;
    int tile_end = tile_begin + tiles_per_chunk;
    tile_end = tile_end < tiles ? tile_end : tiles;
    const int nt = tile_end - tile_begin;
    float* bias_l = reinterpret_cast<float*>(smem + Lds::bias);
    if (tid < 256) bias_l[tid] = bc[tid];

    if (w < 4) {
        // ======================================= matrix waves =======================================
        ox8 wf[2][24];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const OT* row = wc + (size_t)(64 * w + 32 * b + r_) * kFuseIn + 8 * h_;
#pragma unroll
            for (int ks = 0; ks < 24; ++ks) wf[b][ks] = __builtin_bit_cast(ox8, *reinterpret_cast<const u32x4*>(row + 16 * ks));
        }
        // Each matrix wave stores its OWN 64 channels of the out tile (128-byte lines, line-aligned): accumulators -> a wave-private
        // LDS block [32 px][128 B + pad] -> four 16-byte reads per lane in line order -> four buffer stores (8 whole lines each).
        // No other vector-memory operation lives on these waves, so nothing ever waits for the stores; the helper waves lose their
        // store duty (the stamps: 340 - 680 cycles of their ~3 300 per tile, with the matrix waves idle ~1 000 at the barrier).
        const __amdgpu_buffer_rsrc_t osrd = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (size_t)t * HW * kD), 0, HW * kRowBytes, 0x00020000);
        int st_strip = tile_begin / H, st_y = tile_begin - (tile_begin / H) * H;       // tile `it` of the walk (column strips)
        __syncthreads();                                             // P: bias in LDS; the helpers' tile-0 requests visible to each other
        for (int it = 0; it < nt; ++it) {
            K4_STAMP(0, 0);
            __syncthreads();                                         // B(it): operand tile it built
            K4_STAMP(0, 1);
            int r = r_, h = h_;
            asm volatile("" : "+v"(r), "+v"(h));
            const char* at = smem + Lds::atile + (it & 1) * Lds::a_bytes + r * Lds::kARow + h * 16;
            f32x16 acc[2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int g = 0; g < 4; ++g) {                        // accumulator register 4 g + j <-> channel 64 w + 32 b + 8 g + 4 h + j
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_l + 64 * w + 32 * b + 8 * g + 4 * h);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[b][4 * g + j] = b4[j];
                }
            ox8 xf[2][3];
#pragma unroll
            for (int u = 0; u < 3; ++u) xf[0][u] = *reinterpret_cast<const ox8*>(at + 32 * u);
#pragma unroll
            for (int grp = 0; grp < 8; ++grp) {                      // fragments of group grp + 1 requested before the MFMAs of group grp
                if (grp < 7) {
#pragma unroll
                    for (int u = 0; u < 3; ++u) xf[(grp + 1) & 1][u] = *reinterpret_cast<const ox8*>(at + 32 * (3 * (grp + 1) + u));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    acc[0] = mfma16(wf[0][3 * grp + u], xf[grp & 1][u], acc[0]);
                    acc[1] = mfma16(wf[1][3 * grp + u], xf[grp & 1][u], acc[1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            K4_STAMP(0, 2);
            char* ot = smem + Lds::otile + w * Lds::o_wave;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    mx4 o;
#pragma unroll
                    for (int j = 0; j < 4; j += 2) { MT q0, q1; to_map2<MT, BP>(acc[b][4 * g + j], acc[b][4 * g + j + 1], q0, q1); o[j] = q0; o[j + 1] = q1; }
                    *reinterpret_cast<mx4*>(ot + r * Lds::kOWRow + (32 * b + 8 * g + 4 * h) * 2) = o;
                }
            // the wave's own LDS operations complete in order: the read-back sees the writes above
            const int px0 = st_y * W + st_strip * kTilePx;
            const int lane_o = r + 32 * h;
            const int sbase = (px0 + (lane_o >> 3)) * kRowBytes + 128 * w + 16 * (lane_o & 7);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const u32x4 val = *reinterpret_cast<const u32x4*>(ot + (8 * u + (lane_o >> 3)) * Lds::kOWRow + 16 * (lane_o & 7));
                __builtin_amdgcn_raw_buffer_store_b128(val, osrd, sbase + u * 8 * kRowBytes, 0, 0);
            }
            ++st_y;
            if (st_y == H) { st_y = 0; ++st_strip; }
        }
        __syncthreads();                                             // F
        return;
    }

    // =========================================== helper waves ===========================================
    const int hw = w - 4, ht = tid - 256;
    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>((SVPS_LDS const void*)smem);
    u32x4 psrd;
    {
        const uint64_t a = reinterpret_cast<uint64_t>(prev + (size_t)t * Hp * Wp * kD);
        psrd[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
        psrd[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
        psrd[2] = __builtin_amdgcn_readfirstlane((uint32_t)(Hp * Wp) * kRowBytes);
        psrd[3] = 0x00020000u;
    }
    u32x4 csrd;
    {
        const size_t frame = NCHW_F32 ? (size_t)128 * HW * 4 : (size_t)HW * 256;
        const uint64_t a = reinterpret_cast<uint64_t>(static_cast<const char*>(cur_) + (size_t)t * frame);
        csrd[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
        csrd[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
        csrd[2] = __builtin_amdgcn_readfirstlane((uint32_t)frame);
        csrd[3] = 0x00020000u;
    }
    // horizontal blend weights as MFMA B fragments and the transposed tap reads: see v2 (same staging layout); this helper blends
    // channel blocks 2 hw and 2 hw + 1
    mx8 bwx[2];
    {
        const int c0 = (r_ + 1) >> 1;
        const float lx = (r_ & 1) ? 0.25f : 0.75f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kk = 16 * ks + 8 * h_ + j;
                bwx[ks][j] = (MT)(kk == c0 ? 1.f - lx : (kk == c0 + 1 ? lx : 0.f));
            }
    }
    int tap_off[2][2][2];                                            // [block][k-step][half]
    {
        const int gq = lane >> 4, i = lane & 15, q = i >> 2, p4 = i & 3;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int chunk = 4 * (2 * hw + b) + 2 * (gq & 1) + (p4 >> 1), sub = 8 * (p4 & 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    int row = 16 * ks + 8 * (gq >> 1) + q + 4 * half;
                    row = row < Lds::kStageCols ? row : Lds::kStageCols - 1;
                    tap_off[b][ks][half] = row * kRowBytes + ((chunk ^ swz(row)) * 16) + sub;
                }
        }
    }
    struct Geo { int ys0, ys1, xs_base, x0; float wy; };
    constexpr int kTapDma = 5;                                       // tap DMA instructions per helper wave and tile, at most (18 over 4 waves)
    // Taps are shared between consecutive tiles of a strip: output rows 2m+1 and 2m+2 (and 0, 1, 2) blend the SAME two source rows,
    // only the vertical weight differs - a tile whose (strip, first source row) equals its predecessor's re-uses the staged taps
    // (half the tap requests of a tile on average). Returns the staging buffer (0 / 1) that holds the taps of `tile`.
    // Tile geometry without divisions (an integer division is ~40 scalar / vector instructions, and the old code paid five per
    // iteration - the stamps showed 500 cycles of address arithmetic per tile): the request stream walks (strip, row) incrementally.
    struct Req { int buf, px0; float wy; };
    int req_key = -1, req_buf = 1, last_dma = 0;
    int rq_strip = tile_begin / H, rq_y = tile_begin - (tile_begin / H) * H;   // tile 0 of this workgroup
    auto stage_requests = [&](int tile) {
        const bool live = tile < nt;
        Geo g;
        g.x0 = rq_strip * kTilePx;
        g.ys0 = rq_y > 0 ? (rq_y - 1) >> 1 : 0;                      // floor(max((y + 0.5) / 2 - 0.5, 0))
        g.ys1 = g.ys0 + 1 < Hp ? g.ys0 + 1 : Hp - 1;
        g.wy = rq_y == 0 ? 0.f : ((rq_y & 1) ? 0.25f : 0.75f);
        g.xs_base = (g.x0 >> 1) - 1;
        const int px0 = rq_y * W + g.x0;
        const int key = rq_strip * Hp + g.ys0;
        const bool fresh = live && key != req_key;
        if (fresh) {
            req_key = key;
            req_buf ^= 1;
        }
        last_dma = fresh ? (hw < 2 ? 5 : 4) : 0;                     // tap requests this call issues (wave-uniform)
        if (live) {                                                  // next tile of the walk (column strips)
            ++rq_y;
            if (rq_y == H) { rq_y = 0; ++rq_strip; }
        }
#pragma unroll
        for (int k = 0; k < kTapDma; ++k) {
            if (!fresh) break;
            const int q = hw + 4 * k;                                // DMA instruction q (0..17): source row q / 9, staged columns 2 (q % 9), + 1
            if (q >= 18) break;                                      // helpers 2, 3 issue four (wave-uniform: n_req below)
            const int row = q / 9, cp = q - 9 * row;
            int col = g.xs_base + 2 * cp + h_;
            col = col < 0 ? 0 : (col < Wp ? col : Wp - 1);
            const int ys = row ? g.ys1 : g.ys0;
            int voff = (ys * Wp + col) * kRowBytes + (((lane & 31) ^ swz(2 * cp + h_)) * 16);
            const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + Lds::stage + req_buf * Lds::stage_bytes +
                                                                (row * Lds::kStageCols + 2 * cp) * kRowBytes);
            uint32_t keep;
            asm volatile(
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b32 m0, %1\n\t"
                "s_nop 0\n\t"
                "buffer_load_dwordx4 %2, %3, 0 offen lds\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(keep)
                : "s"(dst), "v"(voff), "s"(psrd)
                : "memory");
        }
        return Req{req_buf, px0, g.wy};
    };
    // incoming map of a tile -> registers of the helper threads, consumed one iteration later. The loads are asm with COUNTED waits
    // (wait_map below): as compiler-visible loads hipcc waited for them with a count that does not know the LDS-DMA requests in
    // between, i.e. for most of the PREVIOUS iteration's stores as well (vmcnt retires in order) - a store acknowledgement per tile on
    // the critical path. NCHW fp32: item (channel pair cp, pixel quad pq) = two 16-byte loads; pixel-major bf16: 16 B of a pixel row.
    struct CurRegs { u32x4 v[2][2]; };
    u32x4 crs;
    {
        const size_t frame = NCHW_F32 ? (size_t)128 * HW * 4 : (size_t)HW * 256;
        const uint64_t a = reinterpret_cast<uint64_t>(static_cast<const char*>(cur_) + (size_t)t * frame);
        crs[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
        crs[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
        crs[2] = __builtin_amdgcn_readfirstlane((uint32_t)frame);
        crs[3] = 0x00020000u;
    }
    auto ld16 = [&](int off) {
        u32x4 v;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(off), "s"(crs) : "memory");
        return v;
    };
    auto load_cur = [&](int tile, int px0, CurRegs& cr) {
        // wave-uniform "dropped" bit OR-ed into the offsets, in unsigned arithmetic (a select between two offsets made hipcc issue
        // each load twice under complementary exec masks with s_waitcnt vmcnt(0) in between; frames are below 2 GiB, so bit 31
        // alone puts an offset out of range and nothing added to it can wrap)
        const uint32_t dead = (uint32_t)__builtin_amdgcn_readfirstlane(tile < nt ? 0 : (int)0x80000000u);
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2) {
            const int idx = ht + 256 * j2;
            if constexpr (NCHW_F32) {
                const int cp = idx >> 3, pq = idx & 7;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int off = (int)((uint32_t)(((2 * cp + e) * HW + px0 + 4 * pq) * 4) | dead);
                    cr.v[j2][e] = ld16(off);
                }
            } else {
                const int px = idx >> 4, ck = idx & 15;
                const int off = (int)((uint32_t)((px0 + px) * 256 + ck * 16) | dead);
                cr.v[j2][0] = ld16(off);
            }
        }
    };
    // operand tile of `tile`: incoming map -> channels 256 .. 383 (conversion / copy), blended taps -> channels 0 .. 255
    auto build = [&](int tile, CurRegs& cr, int tap_buf, float wy) {
        char* at = smem + Lds::atile + (tile & 1) * Lds::a_bytes;
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
            for (int e = 0; e < (NCHW_F32 ? 2 : 1); ++e) asm volatile("" : "+v"(cr.v[j2][e]));   // defined by the counted wait above the call
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2) {
            const int idx = ht + 256 * j2;
            if constexpr (NCHW_F32) {
                const int cp = idx >> 3, pq = idx & 7;                 // channels 2cp, 2cp + 1; pixels 4pq .. 4pq + 3
                const f32x4 c0 = __builtin_bit_cast(f32x4, cr.v[j2][0]);
                const f32x4 c1 = __builtin_bit_cast(f32x4, cr.v[j2][1]);
                const int chunk = 32 + (cp >> 2), sub = (cp & 3) * 4;  // 16-byte chunk of channels 256 + 2cp, byte inside it
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    typedef OT bf16x2 __attribute__((ext_vector_type(2)));
                    bf16x2 v2;
                    if constexpr (BP) { v2[0] = (OT)c0[j]; v2[1] = (OT)c1[j]; }
                    else { MT q0, q1; to_map2<MT, false>(c0[j], c1[j], q0, q1); v2[0] = q0; v2[1] = q1; }
                    *reinterpret_cast<bf16x2*>(at + (4 * pq + j) * Lds::kARow + chunk * 16 + sub) = v2;
                }
            } else {
                const int px = idx >> 4, ck = idx & 15;
                *reinterpret_cast<u32x4*>(at + px * Lds::kARow + (32 + ck) * 16) = cr.v[j2][0];
            }
        }
        // bilinear x2 on the matrix cores (v2's arithmetic: each row sum is round(w0 a + w1 b), then h0 top + h1 bottom)
        const float h1 = wy, h0 = 1.f - wy;
        const f32x2 h0v = {h0, h0}, h1v = {h1, h1};
        const int so = Lds::stage + tap_buf * Lds::stage_bytes;
        // both channel blocks of this helper in lock-step (all sixteen transposed reads, then the eight MFMAs, then the vector work):
        // one block after the other was two dependent chains of ~700 cycles each
        mx8 af[2][2][2];                                          // [block][source row][k-step]
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int row = 0; row < 2; ++row)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const char* base = smem + so + row * Lds::kStageCols * kRowBytes;
                    const mx4 lo = ds_tr16((SVPS_LDS mx4*)(base + tap_off[b][ks][0]));
                    const mx4 hi = ds_tr16((SVPS_LDS mx4*)(base + tap_off[b][ks][1]));
                    af[b][row][ks] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
        __builtin_amdgcn_sched_barrier(0);
        f32x16 up[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int row = 0; row < 2; ++row)
#pragma unroll
                for (int i = 0; i < 16; ++i) up[b][row][i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int row = 0; row < 2; ++row)
                    up[b][row] = mfma16(af[b][row][ks], bwx[ks], up[b][row]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int wo = r_ * Lds::kARow + (32 * (2 * hw + b) + 4 * h_) * 2;       // pixel row r_, channels 32 cb + 8 g + 4 h ..
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                ox4 o;
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const f32x2 top = {up[b][0][4 * gq + j], up[b][0][4 * gq + j + 1]}, bot = {up[b][1][4 * gq + j], up[b][1][4 * gq + j + 1]};
                    const f32x2 y = __builtin_elementwise_fma(h1v, bot, h0v * top);
                    if constexpr (BP) { o[j] = (OT)y[0]; o[j + 1] = (OT)y[1]; }
                    else { MT q0, q1; to_map2<MT, false>(y[0], y[1], q0, q1); o[j] = q0; o[j + 1] = q1; }
                }
                *reinterpret_cast<ox4*>(at + wo + 16 * gq) = o;
            }
        }
    };

    constexpr int kCur = NCHW_F32 ? 4 : 2;                           // map loads per helper thread and tile
#ifndef SVPS_K4_AHEAD
#define SVPS_K4_AHEAD 1
#endif
    // the incoming map is loaded kAhead tiles ahead of its use, through kAhead + 1 register sets. Round 4 (the map as 16-bit rows): the
    // stamps show the helpers waiting 560 - 1 400 cycles per tile for loads issued a whole iteration earlier (under the kernel's own
    // store traffic a load takes 3 500 - 4 400 cycles), but two tiles ahead (-DSVPS_K4_AHEAD=2, parity-green) measures the same on one
    // box in both input forms (1 216 - 1 221 against 1 223 - 1 228 us, T = 40, rows; 1 311 - 1 320 against 1 305 - 1 309 NCHW): the wait
    // moves, the memory system's rate for this access mix does not. One ahead stays the shipped schedule
    constexpr int kAhead = (SVPS_K4_AHEAD);
    static_assert(kAhead == 1 || kAhead == 2, "map prefetch depth");
    CurRegs cx, cy, cz;
    int ld_strip = tile_begin / H, ld_y = tile_begin - (tile_begin / H) * H;   // the map loads' own walk over the tiles (column strips)
    auto load_next = [&](int tile, CurRegs& cr) {
        load_cur(tile, ld_y * W + ld_strip * kTilePx, cr);
        if (tile < nt) {
            ++ld_y;
            if (ld_y == H) { ld_y = 0; ++ld_strip; }
        }
    };
    // prologue: tiles 0 and 1 requested (kAhead == 2: the map of tile 2 as well), everything landed (once per workgroup)
    const Req r0 = stage_requests(0);
    load_next(0, cx);
    Req r_next = stage_requests(1);                                  // tile it+1: staging buffer of its taps, first pixel, vertical weight
    load_next(1, cy);
    if constexpr (kAhead == 2) load_next(2, cz);
    wait_vm<0>();
    __syncthreads();                                                 // P: every helper's tap pieces of tiles 0 and 1 visible
    build(0, cx, r0.buf, r0.wy);
    // iteration it: tap requests of tile it+2 (its buffers were consumed by build(it) before B(it)), map loads of tile it+1+kAhead,
    // operand tile it+1 from the registers loaded kAhead iterations ago; issue order taps DMA, map loads
    auto iter = [&](int it, CurRegs& use, CurRegs& load) {
        K4_STAMP(1, 0);
        __syncthreads();                                             // B(it): operand tile it complete; out tile it-1 complete; taps of tile it+1 visible
        K4_STAMP(1, 1);
        const Req r_new = stage_requests(it + 2);                    // (a fresh group's buffer was last read by build(it) before B(it))
        K4_STAMP(1, 2);
        load_next(it + 1 + kAhead, load);
        K4_STAMP(1, 3);
        K4_STAMP(1, 4);
        // the map loads of tile it+1 landed. Younger, in issue order: (kAhead == 2: the map loads of tile it+2, one iteration old,) this
        // iteration's tap requests and map loads
        wait_vm_dyn((kAhead - 1) * kCur + last_dma + kCur);
        K4_STAMP(1, 5);
        if (it + 1 < nt) build(it + 1, use, r_next.buf, r_next.wy);
        K4_STAMP(1, 6);
        r_next = r_new;
        wait_vm_dyn(kCur);                                           // the tap requests of tile it+2 landed (younger: this iteration's map loads)
        K4_STAMP(1, 7);
    };
    if constexpr (kAhead == 1) {
        for (int it = 0; it < nt; it += 2) {
            iter(it, cy, cx);
            if (it + 1 < nt) iter(it + 1, cx, cy);
        }
    } else {                                                         // tile t lives in set t % 3: iteration it builds from set (it+1) % 3, loads into it % 3
        for (int it = 0; it < nt; it += 3) {
            iter(it, cy, cx);
            if (it + 1 < nt) iter(it + 1, cz, cy);
            if (it + 2 < nt) iter(it + 2, cx, cz);
        }
    }
    __syncthreads();                                                 // F
}

}  // namespace svps

namespace {
int fuse_num_cus() { return svps_num_cus(); }

template <typename MT, bool NCHW, bool L0, bool BP = false>
hipError_t launch_fuse(const void* cur, const void* prev, const void* wc, const float* bc, void* out, int T, int H,
                       int W, hipStream_t stream) {
    auto kern = svps::level_fuse_kernel<MT, NCHW, L0, BP>;
    const int HW = H * W;
    const int tiles = (HW + svps::kTilePx - 1) / svps::kTilePx;
    int chunks = svps_pick_chunks(T, tiles, fuse_num_cus());   // one resident 8-wave workgroup per CU
    const int tpc = (tiles + chunks - 1) / chunks;
    chunks = (tiles + tpc - 1) / tpc;
    hipLaunchKernelGGL(kern, dim3(chunks, T), dim3(512), svps::FuseLds::total, stream, cur,
                       static_cast<const MT*>(prev), static_cast<const MT*>(wc), bc,
                       static_cast<MT*>(out), H, W, tpc);
    return hipGetLastError();
}

template <bool NCHW>
hipError_t launch_fuse_v2(const void* cur, const void* prev, const void* wc, const float* bc, void* out, int T, int H,
                          int W, hipStream_t stream) {
    auto kern = svps::level_fuse_kernel_v2<NCHW>;
#ifdef SVPS_K4_ABLATE
    {
        static int abl = -1;
        if (abl < 0) { const char* e = getenv("SVPS_K4_ABLATE"); abl = e ? atoi(e) : 0; }
        switch (abl) {
            case 1: kern = svps::level_fuse_kernel_v2<NCHW, 1>; break;
            case 2: kern = svps::level_fuse_kernel_v2<NCHW, 2>; break;
            case 4: kern = svps::level_fuse_kernel_v2<NCHW, 4>; break;
            case 8: kern = svps::level_fuse_kernel_v2<NCHW, 8>; break;
            case 6: kern = svps::level_fuse_kernel_v2<NCHW, 6>; break;
            case 7: kern = svps::level_fuse_kernel_v2<NCHW, 7>; break;
            case 15: kern = svps::level_fuse_kernel_v2<NCHW, 15>; break;
            case 14: kern = svps::level_fuse_kernel_v2<NCHW, 14>; break;
            case 11: kern = svps::level_fuse_kernel_v2<NCHW, 11>; break;
            default: break;
        }
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, svps::Fuse2Lds::total);
    }
#endif
    static SvpsLdsAttr attr;
    if (hipError_t ae = attr.ensure(reinterpret_cast<const void*>(kern), svps::Fuse2Lds::total); ae != hipSuccess) return ae;
    const int tiles = H * W / svps::kTilePx;
    int chunks = svps_pick_chunks(T, tiles, fuse_num_cus());
    const int tpc = (tiles + chunks - 1) / chunks;
    chunks = (tiles + tpc - 1) / tpc;
    hipLaunchKernelGGL(kern, dim3(chunks, T), dim3(512), svps::Fuse2Lds::total, stream, cur,
                       static_cast<const __bf16*>(prev), static_cast<const __bf16*>(wc), bc,
                       static_cast<__bf16*>(out), H, W, tpc);
    return hipGetLastError();
}
template <typename MT, bool NCHW, bool BP = false>
hipError_t launch_fuse_v4(const void* cur, const void* prev, const void* wc, const float* bc, void* out, int T, int H,
                          int W, hipStream_t stream) {
    auto kern = svps::level_fuse_kernel_v4<MT, NCHW, BP>;
    static SvpsLdsAttr attr;
    if (hipError_t ae = attr.ensure(reinterpret_cast<const void*>(kern), svps::Fuse4Lds::total); ae != hipSuccess) return ae;
    const int tiles = H * W / svps::kTilePx;
    int chunks = svps_pick_chunks(T, tiles, fuse_num_cus());
    const int tpc = (tiles + chunks - 1) / chunks;
    chunks = (tiles + tpc - 1) / tpc;
    hipLaunchKernelGGL(kern, dim3(chunks, T), dim3(512), svps::Fuse4Lds::total, stream, cur,
                       static_cast<const MT*>(prev), wc, bc,
                       static_cast<MT*>(out), H, W, tpc);
    return hipGetLastError();
}
}  // namespace

#ifdef SVPS_K4_STAMP
extern "C" int svps_k4_debug_read(unsigned long long* stamps) {
    return (int)hipMemcpyFromSymbol(stamps, HIP_SYMBOL(svps::k4_stamps), sizeof(unsigned long long) * 2 * 8 * 8);
}
#endif

extern "C" int svps_level_fuse_fwd(const void* cur, int cur_flags, const void* prev, const void* wc,
                                   const float* bc, void* out, int T, int H, int W, void* stream_) {
    if (!cur || !wc || !bc || !out) return SVPS_ERR_BAD_ARG;
    const bool cur_is_nchw_f32 = cur_flags & 1;
    const bool maps_f16 = cur_flags & 2;                       // prev, wc and out are fp16 (three more mantissa bits in the same bytes)
    if (T <= 0 || H <= 0 || W <= 0) return SVPS_ERR_BAD_SHAPE;
    if ((size_t)H * W > svps::kMaxFramePixels) return SVPS_ERR_BAD_SHAPE;   // 32-bit buffer offsets inside a frame
    if (prev && ((H & 1) || (W & 1))) return SVPS_ERR_BAD_SHAPE;   // x2 upsampling: even sizes
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    svps_prof_mark(SVPS_KERNEL_LEVEL_FUSE, 0, stream);
    hipError_t e;
    // fast path: tiles inside one output row, frame sizes inside a buffer descriptor
    static const bool legacy = getenv("SVPS_K4_LEGACY") != nullptr;   // comparison runs: the first-generation kernel
    const bool fast = prev && (W & 31) == 0 && (size_t)H * W * 512 < 0x7fffffffu && !legacy;
    static const bool v2 = getenv("SVPS_K4_V2") != nullptr;          // comparison runs: the eight-wave form of round 2
    const bool bf16_values = cur_flags & 4;                    // fp16 encoding, bf16 rounding points (the bf16 storage policy; see to_map)
    if (maps_f16 && bf16_values && cur_is_nchw_f32)
        e = fast ? launch_fuse_v4<_Float16, true, true>(cur, prev, wc, bc, out, T, H, W, stream)
            : prev ? launch_fuse<_Float16, true, false, true>(cur, prev, wc, bc, out, T, H, W, stream)
                   : launch_fuse<_Float16, true, true, true>(cur, prev, wc, bc, out, T, H, W, stream);
    else if (maps_f16 && bf16_values)                          // pixel-major incoming map: bf16 (the conv's operand type in this form)
        e = fast ? launch_fuse_v4<_Float16, false, true>(cur, prev, wc, bc, out, T, H, W, stream)
            : prev ? launch_fuse<_Float16, false, false, true>(cur, prev, wc, bc, out, T, H, W, stream)
                   : launch_fuse<_Float16, false, true, true>(cur, prev, wc, bc, out, T, H, W, stream);
    else if (maps_f16 && cur_is_nchw_f32)
        e = fast ? launch_fuse_v4<_Float16, true>(cur, prev, wc, bc, out, T, H, W, stream)
            : prev ? launch_fuse<_Float16, true, false>(cur, prev, wc, bc, out, T, H, W, stream)
                   : launch_fuse<_Float16, true, true>(cur, prev, wc, bc, out, T, H, W, stream);
    else if (maps_f16)                                         // pixel-major incoming map: fp16
        e = fast ? launch_fuse_v4<_Float16, false>(cur, prev, wc, bc, out, T, H, W, stream)
            : prev ? launch_fuse<_Float16, false, false>(cur, prev, wc, bc, out, T, H, W, stream)
                   : launch_fuse<_Float16, false, true>(cur, prev, wc, bc, out, T, H, W, stream);
    else if (fast && !v2)
        e = cur_is_nchw_f32 ? launch_fuse_v4<__bf16, true>(cur, prev, wc, bc, out, T, H, W, stream)
                            : launch_fuse_v4<__bf16, false>(cur, prev, wc, bc, out, T, H, W, stream);
    else if (fast)
        e = cur_is_nchw_f32 ? launch_fuse_v2<true>(cur, prev, wc, bc, out, T, H, W, stream)
                            : launch_fuse_v2<false>(cur, prev, wc, bc, out, T, H, W, stream);
    else if (prev)
        e = cur_is_nchw_f32 ? launch_fuse<__bf16, true, false>(cur, prev, wc, bc, out, T, H, W, stream)
                            : launch_fuse<__bf16, false, false>(cur, prev, wc, bc, out, T, H, W, stream);
    else
        e = cur_is_nchw_f32 ? launch_fuse<__bf16, true, true>(cur, prev, wc, bc, out, T, H, W, stream)
                            : launch_fuse<__bf16, false, true>(cur, prev, wc, bc, out, T, H, W, stream);
    svps_prof_mark(SVPS_KERNEL_LEVEL_FUSE, 1, stream);
    return (int)e;
}
