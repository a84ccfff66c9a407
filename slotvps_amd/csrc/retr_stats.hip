// K3' - per-pixel LayerNorm statistics of the key / value projections, for the statistics-fused retriever (gfx950).
//
// MaskDynamicConv.forward (mmdet/models/detectors/dynamic_mask_head.py:428-433) computes, for every pixel p,
//     k_p = norm_k(to_k(f_p + pos_p)),   v_p = norm_v(to_v(f_p))
// and the first form of this library (K3, kv_project.hip) wrote both as bf16 tensors: 1 KiB per pixel and stage out,
// 1 KiB back in by K1, with the 8-bit mantissa of bf16 on every key (1.2e-1 on the slot update against fp32).
// Both LayerNorms are affine in the projection up to ONE scalar per pixel, the reciprocal standard deviation:
//     k_p = gamma_k * rstd_k(p) * (W~_k x_p + b~_k) + beta_k        W~ = (I - 11^T/256) W  (centred rows), x_p = f_p + pos_p
// so the retriever can fold W~_k into the queries and W~_v behind the pixel sum (retr_attn.hip) and needs from the pixel
// side only rstd_k(p), rstd_v(p). This kernel produces them: 16 B per pixel instead of 1024 B.
//
//     var(p) = |W~ x_p + b~|^2 / 256 = |R x_p + r|^2 / 256,      [W~ | b~] = Q [R | r]  (QR, R upper triangular)
// The host factorises once per weight (float64) and hands R as FP16: an upper-triangular 256 x 256 matrix has 36 of 64
// non-zero 32 x 32 blocks, so the statistics cost 56 % of the matrix work of the projection itself. Row blocks are paired
// (j, 7 - j): 9 blocks = 18 v_mfma_f32_32x32x16_f16 per wave and tile, weights resident in 72 VGPRs.
//
// The position term never meets the matrix cores. pos is separable (position_encoding.py:251-255) and R_k is linear:
//     R_k (f_p + pos_p) + r_k = R_k f_p + Ty[y(p)] + Tx[x(p)] + r_k,     Ty = R_k[:, :128] ytab[y],  Tx = R_k[:, 128:] xtab[x]
// two small fp32 tables per (stage, level geometry) that the host side computes once (float64 R_k, no rounding of f + pos).
// They enter as the INITIAL value of the key accumulators. Tiles are 32 consecutive pixels of one image row, walked down a
// 32-pixel-wide column strip (tile id = strip * H + row, the order retr_attn.hip uses), so Tx of a lane's pixel column sits
// in registers for a whole strip and the per-tile position data is one 1-KiB row of Ty, staged by LDS-DMA with the tile.
//
// Output per pixel: ONE 16-byte "aux" row
//     { 1, hi(sigma_v), lo(sigma_v), 0 } FP16, { rstd_k, rstd_v } fp32            sigma_v = 1 / rstd_v
// (a tile's 32 rows = 512 contiguous bytes, whole memory lines). The retriever stages the rows with the value tile and uses the
// four FP16 words as columns of a ninth channel block: with A = P * rstd_v on the matrix cores they accumulate
// s1 = sum_p P rstd_v and s0 = sum_p P (needed for the bias terms) at no vector-ALU cost; its producers read the two fp32 words
// from the staged tile.
//
// Mapping: 8 waves, two per SIMD. Waves 0-3 = key projection, waves 4-7 = value projection; every wave stages two 1-KiB pieces
// of every tile by LDS-DMA and converts them bf16 -> fp16 in place when they have landed (exact for |f| in [6.1e-5, 65504]):
// BOTH sides read the same fp16 tile. Feature tiles five ahead in a 6-deep ring; ping-pong schedule with two workgroup barriers
// per tile (key waves heavy while value waves light, then the reverse: see the loop); the first fragment groups of a tile are
// requested in the light phase before its heavy phase.
// Precision. rstd_k multiplies logits of magnitude up to ~80 in front of a sharp softmax: it has to be good to ~1e-4. The
// operand f is exact, the position term is fp32, R_k / R_v carry 11 bits (fp16). Accumulation and everything after it is fp32.
#include <cstdlib>
#include "common.h"
#include "../../include/slotvps_hip.h"

namespace svps {

#ifndef SVPS_STNT
#define SVPS_STNT 0
#endif
#ifndef SVPS_STA
#define SVPS_STA 3
#endif
constexpr int kStA = SVPS_STA;                // LDS-DMA distance: tile t+2+A is requested in the light phase of tile t
constexpr int kStNF = kStA + 3;          // feature ring depth: tiles t .. t+2+A are live

struct StatsPLds {
    static constexpr int fring = 0;                                       // kStNF x 16 KiB (multiples of 512 B: fragment address XORs)
    static constexpr int tyring = kStNF * kTileBytes;                     // kStNF x 1 KiB: Ty row of the tile's image row
    static constexpr int stats = tyring + kStNF * 1024;                   // [2 tiles][2 proj][4 waves][32 px] float
    static constexpr int total = stats + 2 * 2 * 4 * 32 * 4;
};

__device__ __forceinline__ u32x4 st_make_srd(const void* base, uint32_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    u32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
    d[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
    d[2] = __builtin_amdgcn_readfirstlane(bytes);
    d[3] = 0x00020000u;
    return d;
}

// asm LDS-DMA (see slot_attn.hip: the builtin form makes hipcc drain the ring before every LDS read)
__device__ __forceinline__ void st_dma16(u32x4 srd, uint32_t lds_addr, int voff, int soff) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
#if SVPS_STNT
        "buffer_load_dwordx4 %2, %3, %4 offen nt lds\n\t"
#else
        "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
#endif
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_addr), "v"(voff), "s"(srd), "s"(soff)
        : "memory");
}
__device__ __forceinline__ void st_dma16x4(u32x4 srd, uint32_t lds_addr, int v0, int v1, int v2, int v3, int soff) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %2, %6, %7 offen lds\n\t"
        "buffer_load_dwordx4 %3, %6, %7 offen offset:1024 lds\n\t"
        "buffer_load_dwordx4 %4, %6, %7 offen offset:2048 lds\n\t"
        "buffer_load_dwordx4 %5, %6, %7 offen offset:3072 lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_addr), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(srd), "s"(soff)
        : "memory");
}

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) __fp16 st_fp16x2;

__device__ __forceinline__ float st_half_swap_add(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

#ifdef SVPS_STATS_STAMP
// diagnostic build only (tools/retr_stamps.py): s_memtime stamps of one workgroup's key wave 0 and value wave 0, kept in LDS
// (a global store per stamp would count in vmcnt and turn the counted DMA waits into full drains) and copied out at the end
__device__ unsigned long long stats_stamps[2][8][8];     // [key / value][iteration - 8][point]
#define STATS_STAMP(pt)                                                                               \
    do {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        if (blockIdx.x == 3 && blockIdx.y == 2 && j == 0 && it >= 8 && it < 16 && lane == 0)          \
            reinterpret_cast<unsigned long long*>(smem + StatsPLds::total)[(proj * 8 + (it - 8)) * 8 + pt] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    } while (0)
#define STATS_STAMP_DUMP()                                                                            \
    do {                                                                                              \
        if (blockIdx.x == 3 && blockIdx.y == 2 && j == 0 && lane < 64) {                              \
            const unsigned long long* sl_ = reinterpret_cast<const unsigned long long*>(smem + StatsPLds::total); \
            (&stats_stamps[0][0][0])[proj * 64 + lane] = sl_[proj * 64 + lane];                       \
        }                                                                                             \
    } while (0)
#else
#define STATS_STAMP(pt) do {} while (0)
#define STATS_STAMP_DUMP() do {} while (0)
#endif

// ABL: timing-only ablations (env SVPS_STATS_ABLATE), outputs wrong. 1: no MFMA  2: no conversion  8: no feature DMA after the prologue  32: no wait for the DMA in the light phase
//
// prologue  16: no finish (statistics combine + stores)
template <bool HAS_POS, int PROJ, int J, int ABL = 0>
__device__ __forceinline__ void retr_stats_role(
    const __bf16* __restrict__ feat,    // [T, HW, 256]
    const float* __restrict__ ty,       // [H, 256] R_k[:, :128] ytab[y] or null
    const float* __restrict__ tx,       // [W, 256] R_k[:, 128:] xtab[x] or null
    const _Float16* __restrict__ rk,    // [256, 256] fp16, upper triangular (row = output row of R_k)
    const _Float16* __restrict__ rv,    // [256, 256] fp16, upper triangular
    const float* __restrict__ rbk,      // [256] r_k (the QR-transformed centred bias)
    const float* __restrict__ rbv,
    float eps_k, float eps_v,
    __bf16* __restrict__ aux,           // [T, HW, 8] (16-bit words): 16-byte rows
    int HW, int H, int W, int tiles_per_chunk, int map_f16) {   // map_f16: the map is fp16 already (no conversion in LDS)
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    using Lds = StatsPLds;
    const int lane = threadIdx.x & 63;
    constexpr int proj = PROJ, j = J;            // role of this wave: compile-time, so every fragment index below is static
    const int r = lane & 31, h = lane >> 5;
    const int t = blockIdx.y, c = blockIdx.x;

    const int tiles = ((W + kTilePx - 1) / kTilePx) * H;
    const int tid0 = c * tiles_per_chunk;
    int nt = tiles - tid0;
    nt = nt < tiles_per_chunk ? nt : tiles_per_chunk;
    if (nt <= 0) return;                         // (whole workgroup: the grid never has such a chunk)
    const int strip0 = tid0 / H, row0 = tid0 - strip0 * H;

    // ---- weights: row blocks j (column blocks j..7) and 7 - j (column blocks 7-j..7) as A fragments -----------------
    // block (rb, kb), k-step s in {0, 1}: lane (r, h) holds R[32 rb + r][32 kb + 16 s + 8 h .. + 8]
    constexpr int rb0 = j, rb1 = 7 - j;
    constexpr int NK0 = 2 * (8 - j), NK1 = 2 * (j + 1);       // k-steps of the two row blocks: 18 fragments in all
    f16x8 wf0[NK0], wf1[NK1];
    {
        const _Float16* wsrc = proj ? rv : rk;
#pragma unroll
        for (int i = 0; i < NK0; ++i)
            wf0[i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(wsrc + (size_t)(32 * rb0 + r) * kD + 32 * rb0 + 16 * i + 8 * h));
#pragma unroll
        for (int i = 0; i < NK1; ++i)
            wf1[i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(wsrc + (size_t)(32 * rb1 + r) * kD + 32 * rb1 + 16 * i + 8 * h));
    }
    // constant part of this lane's accumulator rows (acc_row(4g + i, h) = 8g + 4h + i): the bias r, plus - key side - Tx of the
    // lane's pixel column (reloaded when the chunk moves to the next strip)
    // The loads are asm with their own vmcnt(0): a compiler-visible global load inside the main loop makes hipcc wait for
    // vmcnt(0) before the accumulators are initialised in EVERY iteration, which drains the LDS-DMA ring each tile.
    f32x16 b0, b1;
    const u32x4 bsr = st_make_srd(proj ? rbv : rbk, kD * 4u);
    const u32x4 txs = st_make_srd(tx, (HAS_POS && proj == 0) ? (uint32_t)W * 1024u : 0u);
    auto ld16 = [](u32x4 srd, int off) {
        f32x4 v;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(off), "s"(srd) : "memory");
        return v;
    };
    auto load_base = [&](int strip) {
        int xx = kTilePx * strip + r;
        xx = xx < W ? xx : W - 1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 x0 = ld16(bsr, (32 * rb0 + 8 * g + 4 * h) * 4);
            f32x4 x1 = ld16(bsr, (32 * rb1 + 8 * g + 4 * h) * 4);
            if constexpr (HAS_POS && proj == 0) {
                x0 += ld16(txs, (xx * kD + 32 * rb0 + 8 * g + 4 * h) * 4);
                x1 += ld16(txs, (xx * kD + 32 * rb1 + 8 * g + 4 * h) * 4);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { b0[4 * g + i] = x0[i]; b1[4 * g + i] = x1[i]; }
        }
    };
    load_base(strip0);
    wait_vm<0>();

    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>((SVPS_LDS const void*)smem);
    const u32x4 frs = st_make_srd(feat + (size_t)t * HW * kD, (uint32_t)HW * kRowBytes);
    const u32x4 tys = st_make_srd(ty, HAS_POS ? (uint32_t)H * 1024u : 0u);
    // every wave stages (and converts) two DMA pieces = 4 pixel rows of every feature tile; the last wave the Ty row as well
    constexpr int wv = 4 * proj + j;                            // wave number in the workgroup
    int voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 4 * wv + 2 * i + h;
        voff[i] = row * kRowBytes + (((lane & 31) ^ swz(row)) * 16);
    }
    constexpr bool kTyWave = HAS_POS && wv == 7;
    constexpr int nb = 2 + (kTyWave ? 1 : 0);                   // DMA instructions of one batch of this wave
    // stores of one finish() of this wave (they count in vmcnt too): value wave 0 writes the statistics of every tile
    constexpr int nst = ((ABL & 16) || wv != 4) ? 0 : 1;
    int ds = strip0, dy = row0;                                 // strip / image row of the next batch
    auto stage = [&](int tile) {
        if (tile >= nt) return;
        if constexpr (ABL & 8) { if (tile >= kStNF) return; }
        const uint32_t st = __builtin_amdgcn_readfirstlane(lds0 + Lds::fring + (tile % kStNF) * kTileBytes + wv * 2048);
        const int px0 = dy * W + kTilePx * ds;
        const int soff = __builtin_amdgcn_readfirstlane(px0 * kRowBytes);
        if constexpr ((ABL & 128) != 0) {
            if constexpr (proj == 0) {
                st_dma16(frs, st, voff[0], soff);
                st_dma16(frs, st + 1024, voff[1], soff);
                st_dma16(frs, st + 8192, voff[0] + 8192, soff);
                st_dma16(frs, st + 8192 + 1024, voff[1] + 8192, soff);
            }
        } else if (px0 + kTilePx <= HW) {
            st_dma16(frs, st, voff[0], soff);
            st_dma16(frs, st + 1024, voff[1], soff);
        } else {                                       // last row of a ragged strip: clamp the source rows (their results are not stored)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = 4 * wv + 2 * i + h;
                const int src = px0 + row < HW ? row : HW - 1 - px0;
                st_dma16(frs, st + i * 1024, src * kRowBytes + (((lane & 31) ^ swz(row)) * 16), soff);
            }
        }
        if constexpr (kTyWave)
            st_dma16(tys, __builtin_amdgcn_readfirstlane(lds0 + Lds::tyring + (tile % kStNF) * 1024), lane * 16,
                     __builtin_amdgcn_readfirstlane(dy * 1024));
        ++dy;
        if (dy == H) { dy = 0; ++ds; }
    };
    // this wave's two pieces of feature tile `tile`: bf16 -> fp16 in place
    auto convert = [&](int tile) {
        if (tile >= nt || map_f16) return;
        if constexpr (ABL & 2) return;
        const uint32_t st = lds0 + Lds::fring + (tile % kStNF) * kTileBytes + wv * 2048 + lane * 16;
        u32x4 w_[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) w_[i] = *reinterpret_cast<SVPS_LDS const u32x4*>((uintptr_t)(st + i * 1024));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const st_fp16x2 pk = __builtin_amdgcn_cvt_pkrtz(__uint_as_float(w_[i][k] << 16), __uint_as_float(w_[i][k] & 0xffff0000u));
                w_[i][k] = __builtin_bit_cast(uint32_t, pk);
            }
            *reinterpret_cast<SVPS_LDS u32x4*>((uintptr_t)(st + i * 1024)) = w_[i];
        }
    };

    float* stats = reinterpret_cast<float*>(smem + Lds::stats);          // [tile & 1][proj][wave j][px]
    const u32x4 asrd = st_make_srd(aux + (size_t)t * HW * 8, (uint32_t)HW * 16u);

    // fragments: LDS byte address of this lane's 16-B chunk of k-step ks = 8 a + b: (tile + lane_row) ^ (b << 5), + 256 a
    const uint32_t lane_row = lds0 + Lds::fring + r * kRowBytes + ((h ^ swz(r)) << 4);
    // k-steps 2j .. 15 for row block j; row block 7-j joins from k-step 2(7-j). Groups of four k-steps {4g .. 4g+3}, double-
    // buffered: the reads of group g+1 are in flight under the MFMAs of group g; the first group of the NEXT tile is requested
    // before the barrier
    constexpr int G0 = (2 * j) / 4;                              // first group that holds a k-step >= 2j
    constexpr int kXR = 3;
    f16x8 xf[kXR][4];                                            // ring of three groups: group g lives in xf[(g - G0) % 3]
    auto frag = [&](uint32_t tb, int ks) {
        return *reinterpret_cast<SVPS_LDS const f16x8*>((uintptr_t)((tb ^ ((ks & 7) << 5)) + 256 * (ks >> 3)));
    };
    // the first two groups of a tile are requested in the preceding light phase (an LDS read takes 300 - 500 cycles to come
    // back while the partner waves convert and stage)
    auto prefetch = [&](int tile) {
        if constexpr (ABL & 64) return;
        const uint32_t tb = lane_row + (uint32_t)(tile % kStNF) * kTileBytes;
#pragma unroll
        for (int g = G0; g < (G0 + 2 < 4 ? G0 + 2 : 4); ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) xf[(g - G0) % kXR][u] = frag(tb, 4 * g + u);
    };
    int ts = strip0, tyy = row0;                                 // strip / image row of tile `it`

    // sum of squares of (R x + r [+ Ty + Tx]) over this wave's 64 rows, per pixel
    auto heavy = [&](int it) {
        const uint32_t tb = lane_row + (uint32_t)(it % kStNF) * kTileBytes;
        f32x16 a0 = b0, a1 = b1;
        // key side: the Ty row of the tile, added AFTER the MFMAs; requested behind the last fragment group (LDS returns in
        // order: requested first, its eight reads delayed every fragment of the tile)
        f32x4 y0[4], y1[4];
#pragma unroll
        for (int grp = G0; grp < 4; ++grp) {
            if (!(ABL & 64) && grp + 2 < 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) xf[(grp + 2 - G0) % kXR][u] = frag(tb, 4 * (grp + 2) + u);
            }
            if constexpr (HAS_POS && proj == 0) {
                if (grp == 1) {                                  // the iteration that requests the last group (G0 <= 1)
                    const float* tyl = reinterpret_cast<const float*>(smem + Lds::tyring + (it % kStNF) * 1024);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        y0[g] = *reinterpret_cast<const f32x4*>(tyl + 32 * rb0 + 8 * g + 4 * h);
                        y1[g] = *reinterpret_cast<const f32x4*>(tyl + 32 * rb1 + 8 * g + 4 * h);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ks = 4 * grp + u;
                if constexpr (ABL & 1) continue;
                if (ks >= 2 * rb0) a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf0[ks >= 2 * rb0 ? ks - 2 * rb0 : 0], xf[(grp - G0) % kXR][u], a0, 0, 0, 0);
                if (ks >= 2 * rb1) a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf1[ks >= 2 * rb1 ? ks - 2 * rb1 : 0], xf[(grp - G0) % kXR][u], a1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        STATS_STAMP(6);
#ifndef SVPS_TAILPK
#define SVPS_TAILPK 1
#endif
        float tot;
        if constexpr (SVPS_TAILPK != 0) {
            // PACKED fp32 math (v_pk_add_f32 / v_pk_fma_f32: two lanes of work per instruction) is poison beside MFMAs, but here
            // none is in flight: this wave's chain has ended (the first add needs its result) and the partner wave on the SIMD
            // is in its light phase. Half the vector instructions of the tail.
            typedef float f32x2_t __attribute__((ext_vector_type(2)));
            f32x2_t q0 = {0.f, 0.f}, q1 = {0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    f32x2_t u0 = {a0[4 * g + i], a0[4 * g + i + 1]}, u1 = {a1[4 * g + i], a1[4 * g + i + 1]};
                    if constexpr (HAS_POS && proj == 0) {
                        u0 += f32x2_t{y0[g][i], y0[g][i + 1]};
                        u1 += f32x2_t{y1[g][i], y1[g][i + 1]};
                    }
                    q0 = __builtin_elementwise_fma(u0, u0, q0);
                    q1 = __builtin_elementwise_fma(u1, u1, q1);
                }
            tot = st_half_swap_add((q0[0] + q0[1]) + (q1[0] + q1[1]));
        } else {
            if constexpr (HAS_POS && proj == 0) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int i = 0; i < 4; ++i) { a0[4 * g + i] += y0[g][i]; a1[4 * g + i] += y1[g][i]; }
            }
            float s2[4] = {0.f, 0.f, 0.f, 0.f};                       // four independent chains
#pragma unroll
            for (int i = 0; i < 16; ++i) s2[i & 1] = fmaf(a0[i], a0[i], s2[i & 1]);
#pragma unroll
            for (int i = 0; i < 16; ++i) s2[2 + (i & 1)] = fmaf(a1[i], a1[i], s2[2 + (i & 1)]);
            tot = st_half_swap_add((s2[0] + s2[1]) + (s2[2] + s2[3]));
        }
        if (h == 0) stats[(((it & 1) * 2 + proj) * 4 + j) * 32 + r] = tot;
    };

    // statistics of tile `it` -> HBM, by value wave 0 in its light phase: ONE store instruction per tile, 512 contiguous bytes =
    // the tile's 32 aux rows of 16 bytes {1, hi sigma_v, lo sigma_v, 0 (fp16) | rstd_k, rstd_v (fp32)}, whole memory lines only.
    // (Rows written in part - the first layout had 64-byte rows of which 16 were data - make the memory side read each line,
    // merge and write it back; that read-modify-write traffic in the middle of the feature stream cost a third of the kernel's
    // streaming rate: 66 -> 96 us in the DMA-only ablation.)
    int fs = strip0, fy = row0;
    // the eight partial sums of this lane's pixel, requested at the top of the light phase that stores them (the LDS latency runs
    // under the wait for the DMA and the conversion)
    float fin[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto finish_request = [&](int it) {
        if constexpr (wv != 4) return;
        if constexpr (ABL & 16) return;
        const float* spk = stats + ((it & 1) * 2 + 0) * 4 * 32 + r;
        const float* spv = stats + ((it & 1) * 2 + 1) * 4 * 32 + r;
#pragma unroll
        for (int q = 0; q < 4; ++q) { fin[q] = spk[32 * q]; fin[4 + q] = spv[32 * q]; }
    };
    auto finish = [&](int it) {
        const int strip = fs, row = fy;
        if (it >= 0) {
            ++fy;
            if (fy == H) { fy = 0; ++fs; }
        }
        if constexpr (wv != 4) return;
        if constexpr (ABL & 16) return;
        const int xx = kTilePx * strip + r;
        const int px = row * W + xx;
        const bool valid = xx < W && it >= 0;                            // pixels past the right edge of the map: not stored
        const float totk = (fin[0] + fin[1]) + (fin[2] + fin[3]);
        const float totv = (fin[4] + fin[5]) + (fin[6] + fin[7]);
        // v_rsq_f32 (1 ulp) instead of sqrt + divide
        const float vark = totk * (1.f / kD) + eps_k, varv = totv * (1.f / kD) + eps_v;
        const float rstdk = __builtin_amdgcn_rsqf(vark);
        const float rstdv = __builtin_amdgcn_rsqf(varv);
        float sigma = varv * rstdv;
        asm volatile("" : "+v"(sigma));                                          // one fp32 value for both halves (see retr_attn.hip, p2_store)
        const _Float16 sh = (_Float16)sigma;                                     // FP16 hi + lo (K1' runs its value side in fp16)
        const _Float16 sl = (_Float16)(sigma - (float)sh);
        const _Float16 one = (_Float16)1.0f;
        const uint32_t w0 = (uint32_t)__builtin_bit_cast(uint16_t, one) | ((uint32_t)__builtin_bit_cast(uint16_t, sh) << 16);
        const uint32_t w1 = (uint32_t)__builtin_bit_cast(uint16_t, sl);
        const u32x4 row16 = {w0, w1, __float_as_uint(rstdk), __float_as_uint(rstdv)};
        const int aoff = (valid && h == 0) ? px * 16 : 0x7ffffff0;           // out of range -> dropped by the hardware range check
        asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" : : "v"(row16), "v"(aoff), "s"(asrd) : "memory");
    };

    // ---- ping-pong schedule: two barriers per tile ---------------------------------------------------------------------
    //   half-period 2t  : KEY waves   heavy(t)   (fragments + 18 MFMA + sum of squares), alone on their SIMD's matrix pipe
    //                     VALUE waves light(t)   (own pieces of tile t+1 landed -> fp16, request tile t+2+A, first fragments of tile t)
    //   half-period 2t+1: VALUE waves heavy(t);   KEY waves light(t) (own pieces of tile t+2 -> fp16, request tile t+2+A, wave 0:
    //                     statistics of tile t-1 -> HBM, first fragments of tile t+1)
    // A tile is read first by the key waves in half-period 2 * tile: the value waves converted their pieces two half-periods
    // before, the key waves three (one tile earlier than strictly needed, so that nobody converts a tile while a key wave
    // already requests its fragments).
    auto light = [&](int t) {
        [[maybe_unused]] const int it = t;
        if constexpr (proj == 1) finish_request(t - 1);
        constexpr int ahead = proj ? 1 : 2;                      // tile t + ahead: this wave's pieces must be fp16 now
        // landed by now: batch t+ahead - everything except the batches requested after it and the finish() stores issued
        // after it (value wave 0: two per light phase). Steady state: a constant; first / last tiles: everything.
        constexpr int maxy = kStA + 1 - ahead;
        if constexpr (ABL & 32) { if (t + ahead + maxy > nt - 1) wait_vm<0>(); }
        else if (t >= maxy + 1 && t + ahead + maxy <= nt - 1) wait_vm<nb * maxy + nst * (maxy + 1)>();
        else wait_vm<0>();
        STATS_STAMP(4);
        convert(t + ahead);
        STATS_STAMP(5);
        stage(t + 2 + kStA);
        if constexpr (proj == 0) {
            ++tyy;
            if (tyy == H) { tyy = 0; ++ts; if (t + 1 < nt) load_base(ts); }
            if (t + 1 < nt) prefetch(t + 1);
        } else {
            prefetch(t);
            finish(t - 1);                                       // (t == 0: dummy stores outside the data, same vmcnt count)
        }
    };
    // prologue: batches 0 .. A+1 in flight; tile 0 (key waves: and tile 1) landed and converted before the first barrier
#pragma unroll
    for (int b = 0; b < kStA + 2; ++b) stage(b);
    {
        int younger = nt - 1 - (proj ? 0 : 1);
        const int maxy = kStA + 1 - (proj ? 0 : 1);
        younger = younger < 0 ? 0 : (younger > maxy ? maxy : younger);
        wait_vm_dyn(nb * younger);
        convert(0);
        if constexpr (proj == 0) convert(1);
    }
    wg_barrier();
    if constexpr (proj == 0) prefetch(0);
    for (int t = 0; t < nt; ++t) {
        [[maybe_unused]] const int it = t;
        STATS_STAMP(0);
        wg_barrier();                                            // start of half-period 2t
        STATS_STAMP(1);
        if constexpr (proj == 0) heavy(t); else light(t);
        STATS_STAMP(2);
        wg_barrier();                                            // start of half-period 2t+1
        STATS_STAMP(3);
        if constexpr (proj == 0) light(t); else heavy(t);
        STATS_STAMP(7);
    }
    wg_barrier();
    finish_request(nt - 1);
    finish(nt - 1);
    STATS_STAMP_DUMP();
}

template <bool HAS_POS, int ABL = 0>
__global__ __launch_bounds__(512) void retr_stats_kernel(
    const __bf16* __restrict__ feat, const float* __restrict__ ty, const float* __restrict__ tx,
    const _Float16* __restrict__ rk, const _Float16* __restrict__ rv, const float* __restrict__ rbk, const float* __restrict__ rbv,
    float eps_k, float eps_v, __bf16* __restrict__ aux,
    int HW, int H, int W, int tiles_per_chunk, int map_f16) {
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#define SVPS_ROLE(P, JJ) retr_stats_role<HAS_POS, P, JJ, ABL>(feat, ty, tx, rk, rv, rbk, rbv, eps_k, eps_v, aux, HW, H, W, tiles_per_chunk, map_f16)
    switch (w) {                 // every role runs the same sequence of workgroup barriers
        case 0: SVPS_ROLE(0, 0); break;
        case 1: SVPS_ROLE(0, 1); break;
        case 2: SVPS_ROLE(0, 2); break;
        case 3: SVPS_ROLE(0, 3); break;
        case 4: SVPS_ROLE(1, 0); break;
        case 5: SVPS_ROLE(1, 1); break;
        case 6: SVPS_ROLE(1, 2); break;
        default: SVPS_ROLE(1, 3); break;
    }
#undef SVPS_ROLE
}

}  // namespace svps

#ifdef SVPS_STATS_STAMP
static constexpr int kStampLds = 1024;      // stamps of the diagnostic build behind the kernel's own LDS
#else
static constexpr int kStampLds = 0;
#endif
extern "C" int svps_retr_stats_fwd(const void* feat, const float* ty, const float* tx, const void* rk, const float* rbk,
                                   float lnk_eps, const void* rv, const float* rbv, float lnv_eps,
                                   void* aux, int T, int H, int W, int D, int flags, void* stream_) {
    if (!feat || !rk || !rbk || !rv || !rbv || !aux) return SVPS_ERR_BAD_ARG;
    if ((ty == nullptr) != (tx == nullptr)) return SVPS_ERR_BAD_ARG;
    if (D != svps::kD || T <= 0 || H <= 0 || W <= 0) return SVPS_ERR_BAD_SHAPE;
    if ((size_t)H * W > svps::kMaxFramePixels) return SVPS_ERR_BAD_SHAPE;
    const int HW = H * W;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int tiles = ((W + svps::kTilePx - 1) / svps::kTilePx) * H;
    int chunks = svps_pick_chunks(T, tiles, svps_num_cus());
    const int tpc = (tiles + chunks - 1) / chunks;
    chunks = (tiles + tpc - 1) / tpc;
    const bool has_pos = ty != nullptr;
    auto kern = has_pos ? svps::retr_stats_kernel<true> : svps::retr_stats_kernel<false>;
    int slot = has_pos;
#ifdef SVPS_STATS_ABLATE
    static const int abl = [] { const char* e = getenv("SVPS_STATS_ABLATE"); return e ? atoi(e) : 0; }();
    if (has_pos && abl) {
        switch (abl) {
            case 1: kern = svps::retr_stats_kernel<true, 1>; slot = 2; break;
            case 2: kern = svps::retr_stats_kernel<true, 2>; slot = 3; break;
            case 8: kern = svps::retr_stats_kernel<true, 8>; slot = 4; break;
            case 16: kern = svps::retr_stats_kernel<true, 16>; slot = 5; break;
            case 3: kern = svps::retr_stats_kernel<true, 3>; slot = 6; break;
            case 11: kern = svps::retr_stats_kernel<true, 11>; slot = 7; break;
            case 27: kern = svps::retr_stats_kernel<true, 27>; slot = 8; break;
            case 32: kern = svps::retr_stats_kernel<true, 32>; slot = 9; break;
            case 35: kern = svps::retr_stats_kernel<true, 35>; slot = 10; break;
            case 51: kern = svps::retr_stats_kernel<true, 51>; slot = 15; break;
            case 99: kern = svps::retr_stats_kernel<true, 99>; slot = 11; break;
            case 163: kern = svps::retr_stats_kernel<true, 163>; slot = 12; break;
            case 227: kern = svps::retr_stats_kernel<true, 227>; slot = 13; break;
            case 75: kern = svps::retr_stats_kernel<true, 75>; slot = 14; break;
            default: break;
        }
    }
#endif
    static SvpsLdsAttr attr[16];
    if (hipError_t ae = attr[slot].ensure(reinterpret_cast<const void*>(kern), svps::StatsPLds::total + kStampLds); ae != hipSuccess)
        return (int)ae;
    svps_prof_mark(SVPS_KERNEL_RETR_STATS, 0, stream);
    hipLaunchKernelGGL(kern, dim3(chunks, T), dim3(512), svps::StatsPLds::total + kStampLds, stream, static_cast<const __bf16*>(feat),
                       ty, tx, static_cast<const _Float16*>(rk), static_cast<const _Float16*>(rv), rbk, rbv, lnk_eps, lnv_eps,
                       static_cast<__bf16*>(aux), HW, H, W, tpc, (flags & SVPS_FLAG_MAP_F16) ? 1 : 0);
    svps_prof_mark(SVPS_KERNEL_RETR_STATS, 1, stream);
    return (int)hipGetLastError();
}

#ifdef SVPS_STATS_STAMP
extern "C" int svps_stats_debug_read(unsigned long long* stamps) {
    return (int)hipMemcpyFromSymbol(stamps, HIP_SYMBOL(svps::stats_stamps), sizeof(unsigned long long) * 2 * 8 * 8);
}
#endif
