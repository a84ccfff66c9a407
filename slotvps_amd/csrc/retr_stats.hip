// K3' - per-pixel LayerNorm statistics of the key / value projections, for the statistics-fused retriever (gfx950).
//
// MaskDynamicConv.forward (mmdet/models/detectors/dynamic_mask_head.py:428-433) computes, for every pixel p,
//     k_p = norm_k(to_k(f_p + pos_p)),   v_p = norm_v(to_v(f_p))
// and the first form of this library (K3, kv_project.hip) wrote both as bf16 tensors: 1 KiB per pixel and stage out,
// 1 KiB back in by K1, with the 8-bit mantissa of bf16 on every key (1.2e-1 on the slot update against fp32).
// Both LayerNorms are affine in the projection up to ONE scalar per pixel, the reciprocal standard deviation:
//     k_p = gamma_k * rstd_k(p) * (W~_k x_p + b~_k) + beta_k        W~ = (I - 11^T/256) W  (centred rows), x_p = f_p + pos_p
// so the retriever can fold W~_k into the queries and W~_v behind the pixel sum (retr_attn.hip) and needs from the pixel
// side only rstd_k(p), rstd_v(p). This kernel produces them: 8 B per pixel instead of 1024 B.
//
//     var(p) = |W~ x_p + b~|^2 / 256 = |R x_p + r|^2 / 256,      [W~ | b~] = Q [R | r]  (QR, R upper triangular)
// The host factorises once per weight (float64) and hands R as bf16: an upper-triangular 256 x 256 matrix has 36 of 64
// non-zero 32 x 32 blocks, so the statistics cost 56 % of the matrix work of the projection itself. Row blocks are paired
// (j, 7 - j): 9 blocks = 18 v_mfma_f32_32x32x16_bf16 per wave and tile, weights resident in 72 VGPRs.
//
// Output per pixel: rstd_k [T, HW] fp32, rstd_v [T, HW] fp32 and a 64-byte "aux" row of 32 FP16 (+ both rstd as raw fp32)
//     { 1, hi(sigma_v), lo(sigma_v), 0 ... }            sigma_v = 1 / rstd_v
// that the retriever appends to the value tile as a ninth 32-channel block: with A = P * rstd_v on the matrix cores its
// columns accumulate s1 = sum_p P rstd_v and s0 = sum_p P (needed for the bias terms) at no vector-ALU cost.
//
// Mapping: 8 waves, two per SIMD. Waves 0-3 = key projection (operand bf16(f + pos), built in LDS by themselves for the
// NEXT tile, double-buffered), waves 4-7 = value projection (operand = the feature tile as it arrives) and all LDS-DMA
// (feature tiles three ahead in a 4-deep ring, position rows two ahead in a double buffer). One workgroup barrier per tile.
// Precision. rstd_k multiplies logits of magnitude up to ~80 in front of a sharp softmax: it has to be good to ~1e-4, and
// with bf16 operands it is not (measured 3e-4 mean / 1e-3 max relative, mostly the rounding of f + pos). The KEY side
// therefore runs on v_mfma_f32_32x32x16_f16: R_k is handed over as fp16 and the operand is built as fp16(f + pos) - three
// more mantissa bits on both operands at the same matrix rate (f is stored as bf16, so fp16(f) loses nothing unless
// |f + pos| >= 65504, which overflows to inf and shows up as NaN downstream - fused maps are O(10)). rstd_v only scales the
// contribution of its own pixel to a sum over thousands of pixels; the VALUE side stays bf16 x bf16 on the tile as it arrives
// (1e-4 mean / 4e-4 max). Accumulation and everything after it is fp32.
#include <cstdlib>
#include "common.h"
#include "../../include/slotvps_hip.h"

namespace svps {

constexpr int kStNF = 4;                 // feature ring depth: f(it+3) is requested in iteration it

struct StatsPLds {
    static constexpr int fring = 0;                                       // kStNF x 16 KiB
    static constexpr int kXkRow = kRowBytes + 16;                          // key operand rows padded to 528 B (conflict-free, no swizzle)
    static constexpr int kXkTile = kTilePx * kXkRow;
    static constexpr int xk = kStNF * kTileBytes;                          // [2] key operand tiles
    static constexpr int posx = xk + 2 * kXkTile;                          // [2][32 px][128] fp32 xtab rows (unaligned: [0] = ytab rows, [1] = xtab rows)
    static constexpr int posy = posx + 2 * kTileBytes;                     // [2][256] fp32 (aligned tiles only)
    static constexpr int posx_of(bool aligned, int tile) { return posx + (aligned ? (tile & 1) * kTileBytes : kTileBytes); }
    static constexpr int posy_of(bool aligned, int tile) { return aligned ? posy + (tile & 1) * 1024 : posx; }
    static constexpr int stats = posy + 2048;                              // [2 tiles][2 proj][4 waves][32 px] float
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
        "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
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

__device__ __forceinline__ float st_half_swap_add(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// ABL: timing-only ablations (env SVPS_STATS_ABLATE), outputs wrong. 1: no MFMA  2: no build of the key operand
// 4: no position DMA  8: no feature DMA after the prologue  16: no finish (statistics combine + stores)
template <bool HAS_POS, int PROJ, int J, int ABL = 0>
__device__ __forceinline__ void retr_stats_role(
    const __bf16* __restrict__ feat,    // [T, HW, 256]
    const float* __restrict__ pos_y,    // [H, 128] or null
    const float* __restrict__ pos_x,    // [W, 128] or null
    const __bf16* __restrict__ rk,      // [256, 256] FP16 bits, upper triangular (row = output row of R_k)
    const __bf16* __restrict__ rv,      // [256, 256] bf16, upper triangular
    const float* __restrict__ rbk,      // [256] r_k (the QR-transformed centred bias)
    const float* __restrict__ rbv,
    float eps_k, float eps_v,
    float* __restrict__ rstd_k,         // [T, HW]
    float* __restrict__ rstd_v,         // [T, HW]
    __bf16* __restrict__ aux,           // [T, HW, 32]
    int HW, int W, int tiles_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using Lds = StatsPLds;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    constexpr int proj = PROJ, j = J;            // role of this wave: compile-time, so every fragment index below is static
    const int r = lane & 31, h = lane >> 5;
    const int t = blockIdx.y, c = blockIdx.x;

    const int px_begin = c * tiles_per_chunk * kTilePx;
    int px_end = px_begin + tiles_per_chunk * kTilePx;
    px_end = px_end < HW ? px_end : HW;
    const int nt = (px_end - px_begin + kTilePx - 1) / kTilePx;
    if (nt <= 0) return;

    // ---- weights: row blocks j (column blocks j..7) and 7 - j (column blocks 7-j..7) as A fragments -----------------
    // block (rb, kb), k-step s in {0, 1}: lane (r, h) holds R[32 rb + r][32 kb + 16 s + 8 h .. + 8]
    constexpr int rb0 = j, rb1 = 7 - j;
    constexpr int NK0 = 2 * (8 - j), NK1 = 2 * (j + 1);       // k-steps of the two row blocks: 18 fragments in all
    bf16x8 wf0[NK0], wf1[NK1];
    {
        const __bf16* wsrc = proj ? rv : rk;
#pragma unroll
        for (int i = 0; i < NK0; ++i)
            wf0[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wsrc + (size_t)(32 * rb0 + r) * kD + 32 * rb0 + 16 * i + 8 * h));
#pragma unroll
        for (int i = 0; i < NK1; ++i)
            wf1[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wsrc + (size_t)(32 * rb1 + r) * kD + 32 * rb1 + 16 * i + 8 * h));
    }
    // bias of this lane's accumulator rows (acc_row(4g + i, h) = 8g + 4h + i)
    f32x16 b0, b1;
    {
        const float* bsrc = proj ? rbv : rbk;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(bsrc + 32 * rb0 + 8 * g + 4 * h);
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(bsrc + 32 * rb1 + 8 * g + 4 * h);
#pragma unroll
            for (int i = 0; i < 4; ++i) { b0[4 * g + i] = x0[i]; b1[4 * g + i] = x1[i]; }
        }
    }
    wait_vm<0>();

    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>((SVPS_LDS const void*)smem);
    const u32x4 frs = st_make_srd(feat + (size_t)t * HW * kD, (uint32_t)HW * kRowBytes);
    // value wave j: four DMA pieces (8 pixel rows) of every feature tile
    int voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * j + 2 * i + h;
        voff[i] = row * kRowBytes + (((lane & 31) ^ swz(row)) * 16);
    }
    auto stage_f = [&](int tile) {
        if (tile >= nt) return;
        if constexpr (ABL & 8) { if (tile >= kStNF) return; }
        const uint32_t st = __builtin_amdgcn_readfirstlane(lds0 + Lds::fring + (tile % kStNF) * kTileBytes + j * 4096);
        const int px0 = px_begin + tile * kTilePx;
        const int soff = __builtin_amdgcn_readfirstlane(px0 * kRowBytes);
        if (px0 + kTilePx <= HW) {
            st_dma16x4(frs, st, voff[0], voff[1] - 1024, voff[2] - 2048, voff[3] - 3072, soff);
        } else {                                       // ragged last tile of the frame: clamp the source rows
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 8 * j + 2 * i + h;
                const int src = px0 + row < HW ? row : HW - 1 - px0;
                st_dma16(frs, st + i * 1024, src * kRowBytes + (((lane & 31) ^ swz(row)) * 16), soff);
            }
        }
    };
    const u32x4 ysrd = st_make_srd(pos_y, HAS_POS ? (uint32_t)((HW + W - 1) / W) * 512u : 0u);
    const u32x4 xsrd = st_make_srd(pos_x, HAS_POS ? (uint32_t)W * 512u : 0u);
    const bool aligned_rows = (W & 31) == 0;
    // position rows of tile `tile`, by the value waves (always the same number of DMA instructions per wave: the counted
    // waits below rely on it)
    auto stage_pos = [&](int tile) {
        if constexpr (!HAS_POS) return;
        if constexpr (ABL & 4) return;
        if (tile >= nt) return;
        const uint32_t sy = __builtin_amdgcn_readfirstlane(lds0 + Lds::posy_of(aligned_rows, tile) + j * 4096);
        const uint32_t sx = __builtin_amdgcn_readfirstlane(lds0 + Lds::posx_of(aligned_rows, tile) + j * 4096);
        if (aligned_rows) {
            const int px0 = px_begin + tile * kTilePx;
            const int y0 = __builtin_amdgcn_readfirstlane(px0 / W), x0 = px0 - y0 * W;
            if (j == 0) st_dma16(ysrd, __builtin_amdgcn_readfirstlane(lds0 + Lds::posy_of(true, tile)), lane * 16, y0 * 512);
            const int v = j * 4096 + lane * 16;
            st_dma16x4(xsrd, sx, v, v, v, v, __builtin_amdgcn_readfirstlane(x0 * 512));
            return;
        }
        const int p = px_begin + tile * kTilePx + 8 * j + h;
        const int yy = p / W, xx = p - yy * W;                 // rows past the image read zeros (buffer bounds)
        int yo = yy * 512 + (lane & 31) * 16, xo = xx * 512 + (lane & 31) * 16;
        const int xwrap = W * 512 + (lane & 31) * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            st_dma16(ysrd, sy + i * 1024, yo, 0);
            st_dma16(xsrd, sx + i * 1024, xo, 0);
            xo += 1024;                                        // two pixels on; at most two row wraps (W == 1)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const bool wrap = xo >= xwrap;
                xo = wrap ? xo - W * 512 : xo;
                yo = wrap ? yo + 512 : yo;
            }
        }
    };
    const int pa = (aligned_rows || !HAS_POS) ? 2 : 1;                   // position rows: tiles ahead
    const int npos = HAS_POS ? (aligned_rows ? (j == 0 ? 5 : 4) : 8) : 0;

    // xk(tile) = bf16(f(tile) + pos(tile)) by the key waves, LDS only: thread -> 16-byte chunk cpos of pixel rows q + 8u
    auto build_xk = [&](int tile) {
        if constexpr (ABL & 2) return;
        int lt = tid;
        asm volatile("" : "+v"(lt));
        const int q = lt >> 5, cpos = lt & 31;
        const int fe = Lds::fring + (tile % kStNF) * kTileBytes + q * kRowBytes + ((cpos ^ swz(q)) << 4);
        const int fo = fe ^ 32;                                 // swz(8u + q) = swz(q) ^ (2 if u is odd)
        const bool ypart = cpos < 16;
        const int pb = (ypart ? Lds::posy_of(aligned_rows, tile) + (aligned_rows ? 0 : q * 512)
                              : Lds::posx_of(aligned_rows, tile) + q * 512) + (cpos & 15) * 32;
        const int ps = (ypart && aligned_rows) ? 0 : 8 * 512;
        const int xo = Lds::xk + (tile & 1) * Lds::kXkTile + q * Lds::kXkRow + cpos * 16;
        bf16x8 fv[4];
        f32x4 pv[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            fv[u] = *reinterpret_cast<const bf16x8*>(smem + ((u & 1) ? fo : fe) + u * 8 * kRowBytes);
            if constexpr (HAS_POS) {
                const char* pt = smem + pb + u * ps;
                pv[u][0] = *reinterpret_cast<const f32x4*>(pt);
                pv[u][1] = *reinterpret_cast<const f32x4*>(pt + 16);
            } else {
                pv[u][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                pv[u][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f16x8 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o[i] = (_Float16)((float)fv[u][i] + pv[u][0][i]);
                o[4 + i] = (_Float16)((float)fv[u][4 + i] + pv[u][1][i]);
            }
            *reinterpret_cast<f16x8*>(smem + xo + u * 8 * Lds::kXkRow) = o;
        }
    };

    float* stats = reinterpret_cast<float*>(smem + Lds::stats);          // [tile & 1][proj][wave j][px]
    const float eps = proj ? eps_v : eps_k;
    const u32x4 ksrd = st_make_srd(rstd_k + (size_t)t * HW, (uint32_t)px_end * 4u);
    const u32x4 vsrd = st_make_srd(rstd_v + (size_t)t * HW, (uint32_t)px_end * 4u);
    const u32x4 asrd = st_make_srd(aux + (size_t)t * HW * 32, (uint32_t)px_end * 64u);

    // sum of squares of (R x + r) over this wave's 64 rows, per pixel
    auto heavy = [&](int it) {
        constexpr bool padded = proj == 0;                     // key operand: the fp16 tile built by build_xk (padded rows)
        const char* bt = padded ? smem + Lds::xk + (it & 1) * Lds::kXkTile : smem + Lds::fring + (it % kStNF) * kTileBytes;
        int rr = r, hh = h;
        asm volatile("" : "+v"(rr), "+v"(hh));
        f32x16 a0 = b0, a1 = b1;
        int o8[8];
        const int s4 = swz(rr);
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8)
            o8[k8] = padded ? rr * Lds::kXkRow + ((2 * k8 + hh) << 4) : rr * kRowBytes + (((2 * k8 + hh) ^ s4) << 4);
        auto frag = [&](int ks) { return *reinterpret_cast<const bf16x8*>(bt + o8[ks & 7] + (ks >> 3) * 256); };
        // k-steps 2j .. 15 for row block j; row block 7-j joins from k-step 2(7-j). Fragments in groups of four, double-
        // buffered: the reads of group g+1 are in flight under the MFMAs of group g (no exposed LDS latency after the first)
        constexpr int G0 = (2 * j) / 4;                          // first group that holds a k-step >= 2j
        bf16x8 xf[2][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) xf[G0 & 1][u] = frag(4 * G0 + u);
#pragma unroll
        for (int grp = G0; grp < 4; ++grp) {
            if (grp < 3) {
#pragma unroll
                for (int u = 0; u < 4; ++u) xf[(grp + 1) & 1][u] = frag(4 * (grp + 1) + u);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ks = 4 * grp + u;
                if constexpr (ABL & 1) continue;
                if constexpr (proj == 0) {                      // key side: fp16 x fp16
                    const f16x8 xh = __builtin_bit_cast(f16x8, xf[grp & 1][u]);
                    if (ks >= 2 * rb0) a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf0[ks >= 2 * rb0 ? ks - 2 * rb0 : 0]), xh, a0, 0, 0, 0);
                    if (ks >= 2 * rb1) a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf1[ks >= 2 * rb1 ? ks - 2 * rb1 : 0]), xh, a1, 0, 0, 0);
                } else {
                    if (ks >= 2 * rb0) a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf0[ks >= 2 * rb0 ? ks - 2 * rb0 : 0], xf[grp & 1][u], a0, 0, 0, 0);
                    if (ks >= 2 * rb1) a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf1[ks >= 2 * rb1 ? ks - 2 * rb1 : 0], xf[grp & 1][u], a1, 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s2 = fmaf(a0[i], a0[i], s2);
#pragma unroll
        for (int i = 0; i < 16; ++i) s2 = fmaf(a1[i], a1[i], s2);
        s2 = st_half_swap_add(s2);
        if (hh == 0) stats[(((it & 1) * 2 + proj) * 4 + j) * 32 + rr] = s2;
    };

    // statistics of tile `it` -> HBM, by wave 0 of each projection (after the barrier that follows heavy(it))
    auto finish = [&](int it) {
        if (j != 0) return;
        if constexpr (ABL & 16) return;
        const float* sp = stats + ((it & 1) * 2 + proj) * 4 * 32 + r;
        const float tot = (sp[0] + sp[32]) + (sp[64] + sp[96]);
        const float sigma = sqrtf(tot * (1.f / kD) + eps);
        const float rstd = 1.f / sigma;
        const int px = px_begin + it * kTilePx + r;
        const bool mine = h == 0 && px < px_end;
        const int off = mine ? px * 4 : 0x7ffffff0;                      // out of range -> dropped by the hardware range check
        asm volatile("buffer_store_dword %0, %1, %2, 0 offen" : : "v"(rstd), "v"(off), "s"(proj ? vsrd : ksrd) : "memory");
        if (proj) {
            // aux row (FP16): lanes h == 0 store bytes [0, 32) = {1, hi, lo, 0 x 5 | rstd_k, rstd_v as raw fp32, 0 x 4}, lanes h == 1
            // bytes [32, 64) = zeros. Columns 0 .. 2 are the ninth channel block of K1' (s1 / s0 sums); the two fp32 words in
            // columns 8 .. 11 are what K1's producers read from the staged tile instead of two more global loads per tile
            // (as 16-bit columns they are garbage that only reaches accumulator columns nobody stores).
            const _Float16 sh = (_Float16)sigma;                                 // FP16 hi + lo (K1' runs its value side in fp16)
            const _Float16 sl = (_Float16)(sigma - (float)sh);
            const _Float16 one = (_Float16)1.0f;
            const uint32_t w0 = (uint32_t)__builtin_bit_cast(uint16_t, one) | ((uint32_t)__builtin_bit_cast(uint16_t, sh) << 16);
            const uint32_t w1 = (uint32_t)__builtin_bit_cast(uint16_t, sl);
            const float* spk = stats + ((it & 1) * 2 + 0) * 4 * 32 + r;          // the key side's sums of the same tile
            const float totk = (spk[0] + spk[32]) + (spk[64] + spk[96]);
            const float rstdk = 1.f / sqrtf(totk * (1.f / kD) + eps_k);          // bit-identical to the key wave's own value
            u32x4 lo = {h == 0 ? w0 : 0u, h == 0 ? w1 : 0u, 0u, 0u};
            const u32x4 z = {h == 0 ? __float_as_uint(rstdk) : 0u, h == 0 ? __float_as_uint(rstd) : 0u, 0u, 0u};
            const int aoff = px < px_end ? px * 64 + h * 32 : 0x7ffffff0;
            asm volatile("buffer_store_dwordx4 %0, %2, %3, 0 offen\n\ts_nop 1\n\t"
                         "buffer_store_dwordx4 %1, %2, %3, 0 offen offset:16\n\ts_nop 1"
                         : : "v"(lo), "v"(z), "v"(aoff), "s"(asrd) : "memory");
        }
    };

    // ---- prologue ---------------------------------------------------------------------------------------------------
    if (proj) {
        stage_pos(0);
        if (pa == 2) stage_pos(1);
        stage_f(0);
        stage_f(1);
        stage_f(2);
        wait_vm_dyn(4 * ((1 < nt) + (2 < nt)));                 // f(0) and the position rows before it have landed
    }
    wg_barrier();
    if (!proj) build_xk(0);

    // ---- main loop: one barrier per tile --------------------------------------------------------------------------
    //   B(it): xk(it) is built; f(it), f(it+1), pos(it+1) have landed; statistics of tile it-1 are in LDS
    for (int it = 0; it < nt; ++it) {
        if (proj) {
            // landed by now: everything except the feature tile requested last (f(it+2), issued in iteration it-1)
            if constexpr (ABL & 8) wait_vm<0>();
            else wait_vm_dyn(it + 2 < nt ? 4 : 0);
        }
        wg_barrier();
        if (it >= 1) finish(it - 1);
        if (proj) {
            if (pa == 2) stage_pos(it + 2);
            else stage_pos(it + 1);
            stage_f(it + 3);
            if (pa == 1) {                                      // position rows requested one tile ahead: wait for them now
                wait_vm_dyn(it + 3 < nt ? 4 : 0);               // (unaligned widths: the small levels and the tests)
            }
            heavy(it);
        } else {
            heavy(it);
        }
        if (!proj && it + 1 < nt) {
            if (pa == 1) wg_barrier();                          // unaligned: pos(it+1) was requested in this iteration
            build_xk(it + 1);
        } else if (pa == 1 && it + 1 < nt) {
            wg_barrier();
        }
    }
    wg_barrier();
    finish(nt - 1);
    (void)npos;
}

template <bool HAS_POS, int ABL = 0>
__global__ __launch_bounds__(512) void retr_stats_kernel(
    const __bf16* __restrict__ feat, const float* __restrict__ pos_y, const float* __restrict__ pos_x,
    const __bf16* __restrict__ rk, const __bf16* __restrict__ rv, const float* __restrict__ rbk, const float* __restrict__ rbv,
    float eps_k, float eps_v, float* __restrict__ rstd_k, float* __restrict__ rstd_v, __bf16* __restrict__ aux,
    int HW, int W, int tiles_per_chunk) {
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#define SVPS_ROLE(P, JJ) retr_stats_role<HAS_POS, P, JJ, ABL>(feat, pos_y, pos_x, rk, rv, rbk, rbv, eps_k, eps_v, rstd_k, rstd_v, aux, HW, W, tiles_per_chunk)
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

extern "C" int svps_retr_stats_fwd(const void* feat, const float* pos_y, const float* pos_x, const void* rk, const float* rbk,
                                   float lnk_eps, const void* rv, const float* rbv, float lnv_eps, float* rstd_k, float* rstd_v,
                                   void* aux, int T, int H, int W, int D, void* stream_) {
    if (!feat || !rk || !rbk || !rv || !rbv || !rstd_k || !rstd_v || !aux) return SVPS_ERR_BAD_ARG;
    if ((pos_y == nullptr) != (pos_x == nullptr)) return SVPS_ERR_BAD_ARG;
    if (D != svps::kD || T <= 0 || H <= 0 || W <= 0) return SVPS_ERR_BAD_SHAPE;
    if ((size_t)H * W > svps::kMaxFramePixels) return SVPS_ERR_BAD_SHAPE;
    const int HW = H * W;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int tiles = (HW + svps::kTilePx - 1) / svps::kTilePx;
    int chunks = svps_pick_chunks(T, tiles, svps_num_cus());
    const int tpc = (tiles + chunks - 1) / chunks;
    chunks = (tiles + tpc - 1) / tpc;
    const bool has_pos = pos_y != nullptr;
    auto kern = has_pos ? svps::retr_stats_kernel<true> : svps::retr_stats_kernel<false>;
    int slot = has_pos;
#ifdef SVPS_STATS_ABLATE
    static const int abl = [] { const char* e = getenv("SVPS_STATS_ABLATE"); return e ? atoi(e) : 0; }();
    if (has_pos && abl) {
        switch (abl) {
            case 1: kern = svps::retr_stats_kernel<true, 1>; slot = 2; break;
            case 2: kern = svps::retr_stats_kernel<true, 2>; slot = 3; break;
            case 4: kern = svps::retr_stats_kernel<true, 4>; slot = 4; break;
            case 6: kern = svps::retr_stats_kernel<true, 6>; slot = 5; break;
            case 8: kern = svps::retr_stats_kernel<true, 8>; slot = 6; break;
            case 16: kern = svps::retr_stats_kernel<true, 16>; slot = 7; break;
            case 7: kern = svps::retr_stats_kernel<true, 7>; slot = 8; break;
            case 14: kern = svps::retr_stats_kernel<true, 14>; slot = 9; break;
            default: break;
        }
    }
#endif
    static SvpsLdsAttr attr[10];
    if (hipError_t ae = attr[slot].ensure(reinterpret_cast<const void*>(kern), svps::StatsPLds::total); ae != hipSuccess)
        return (int)ae;
    svps_prof_mark(SVPS_KERNEL_RETR_STATS, 0, stream);
    hipLaunchKernelGGL(kern, dim3(chunks, T), dim3(512), svps::StatsPLds::total, stream, static_cast<const __bf16*>(feat),
                       pos_y, pos_x, static_cast<const __bf16*>(rk), static_cast<const __bf16*>(rv), rbk, rbv, lnk_eps, lnv_eps,
                       rstd_k, rstd_v, static_cast<__bf16*>(aux), HW, W, tpc);
    svps_prof_mark(SVPS_KERNEL_RETR_STATS, 1, stream);
    return (int)hipGetLastError();
}
