// K1' - statistics-fused slot <-> pixel retriever for gfx950: reads the fused feature map ONCE per stage, no k / v tensors.
//
// MaskDynamicConv.forward (mmdet/models/detectors/dynamic_mask_head.py:423-461), with both pixel-side LayerNorms written
// as "one scalar per pixel times an affine map" (see retr_stats.hip for rstd_k, rstd_v and the centred weights W~, b~):
//
//   logits (:435)   S[l, p] = q_l . k_p = rstd_k(p) * ( Q''_l . f_p + Q''_l . pos_p + a'_l ) + c3_l
//                       Q''_l = W~_k^T (q_l * gamma_k)   a'_l = (q_l * gamma_k) . b~_k   c3_l = q_l . beta_k
//                   pos is separable (position_encoding.py:251-255): Q''_l . pos_p = Cy[y(p), l] + Cx[x(p), l], two small
//                   fp32 tables per frame and stage (a' folded into Cy) - the position term never passes through bf16
//   softmax (:446)  over the SLOT axis, per pixel (no cross-tile state)
//   attn.v  (:456)  o_l = sum_p P[l, p] v_p = gamma_v * ( W~_v A_l + b~_v s1_l ) + beta_v s0_l
//                       A_l = sum_p P rstd_v f_p    s1_l = sum_p P rstd_v    s0_l = sum_p P
//                   this kernel accumulates A (and s1, s0 through a ninth "aux" channel block, retr_stats.hip); the
//                   256 x 256 product with W~_v, norm1 and ReLU (:458-459) follow on [L, 256] tensors (slot side).
//
// Q'' is carried as FP16 hi + lo (22-bit mantissa), P * rstd_v as ONE FP16, f is the stored bf16 map converted to FP16 in LDS
// (exact): the only roundings are those, fp32 accumulation and v_exp_f32. Against a float64 evaluation of the reference formulas
// on the same bf16 map the slot update agrees to ~1e-3 (bf16 k / v: 1.2e-1).
//
// Structure: 8 waves, producers 0-3 / consumers 4-7, one producer and one consumer per SIMD, ONE barrier per tile, LDS-DMA ring
// three tiles ahead (details above retr_attn_kernel).
//   producer sb: logits of slot block sb: 16 + 16 MFMA (Q'' hi, lo) on the row fragments of f(t), started from Cy + Cx;
//       * rstd_k + c3; softmax statistics exchange; P * rstd_v -> LDS (the finish of tile t-1 in the shadow of the chain of tile t)
//   consumer sb: A[sb, 0:256] += P(t-2) f(t-2) and the aux block: 18 MFMA per tile; all LDS-DMA; bf16 -> fp16 of its pieces
// LDS: feature ring 6 x 16 KiB, aux ring 6 x 1 KiB (16-byte rows of retr_stats.hip), Cy ring 6 x 1 KiB, P ring 2 x 8 KiB.
// Measured (finest level, T = 5): 124 - 131 us depending on the box (first version 184); DMA + barriers alone 57 us (the per-CU
// LDS-DMA fill rate); DESIGN.md 3 and 7 carry the stamps, ablations and what was tried.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "retr_common.h"
#include "../../include/slotvps_hip.h"

namespace svps {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(2))) __fp16 fp16x2_t;
typedef __fp16 fp16x4_gcc __attribute__((__vector_size__(4 * sizeof(__fp16))));

#ifndef SVPS_RETR_HL_PREFETCH
#define SVPS_RETR_HL_PREFETCH 4     // HL form: tiles requested ahead (4 fits the LDS since the sixteen-row P tiles of round 6)
#endif
#ifndef SVPS_RETR_PREFETCH
#define SVPS_RETR_PREFETCH 3        // tiles requested ahead (tools/variants.sh retr_attn SVPS_RETR_PREFETCH 2 3 4)
#endif
constexpr int kRPrefetch = SVPS_RETR_PREFETCH;
constexpr int kRNF = kRPrefetch + 3;       // feature / aux / Cy ring depth: tiles it-2 .. it+3 are live in iteration it
template <int PT>    // PT = 1: P * rstd_v as one fp16 tile; 2: hi and lo tiles (the precision form, retr_attn_kernel<.., PHL = true>)
struct RetrLdsT {
    static constexpr int kA = kRPrefetch;
    static constexpr int kNF = kRNF;
    static constexpr int kPBufBytes = PT * kPTile;              // one P buffer: the hi tile (and the lo tile behind it)
    static constexpr int kPLo = kPTile;
    static constexpr int fring = 0;                             // tile bases are multiples of 512 B (fragment address XORs)
    static constexpr int aring = kRNF * kTileBytes;
    static constexpr int yring = aring + kRNF * kAuxTile;
    static constexpr int pring = yring + kRNF * kCyTile;        // [2][8 KiB] fp16, slot block sb at sb * 2 KiB (32 pixel rows of 64 B)
    static constexpr int stats = pring + 2 * PT * kPTile;       // [2][4][32] float2
    static constexpr int c3 = stats + 2 * 4 * 32 * 8;           // [128] float
    static constexpr int total = c3 + 128 * 4;
};
using RetrLds = RetrLdsT<1>;
static_assert(RetrLds::pring % 512 == 0 && RetrLds::total <= 160 * 1024 && RetrLdsT<2>::total <= 160 * 1024, "LDS layout");
// HL form (round 6): a tile is sixteen pixels, so a P tile has sixteen rows per slot block - the hi rows at sb * 2 KiB + row * 64, the lo
// rows in the other half of the block (+ 1 KiB): ONE 8-KiB tile per buffer instead of two with duplicated rows (round 5 wrote every row
// twice so that the consumers' lo k-step could read rows 16 .. 31: it now re-uses the fragment of k-step 0). The 16 KiB this frees were
// tried as a SEVENTH ring stage for a pipelined producer schedule (the softmax head of tile it-1 in the shadow of chain(it), the consumers
// three tiles behind): correct, and 3.7 % SLOWER on the same box (2 165 against 2 090 us, finest level, T = 40) - what bounds the
// kernel is the matrix pipe shared by a producer and a consumer wave per SIMD (58 MFMAs x 32 cycles = 1 856 of ~2 840 cycles per tile), not the
// head's latency; archived as profiles/r06/k1hl_pipelined_head_experiment.patch with its stamps (profiles/r06/README.md).
struct RetrLdsHL {
    static constexpr int kA = SVPS_RETR_HL_PREFETCH;            // batches requested ahead
    static constexpr int kNF = SVPS_RETR_HL_PREFETCH + 3;       // tiles it-2 .. it+A are live in iteration it
    static constexpr int kPBufBytes = kPTile;
    static constexpr int kPLo = 1024;                           // lo rows of a slot block: behind its sixteen hi rows
    static constexpr int fring = 0;
    static constexpr int aring = kNF * kTileBytes;
    static constexpr int yring = aring + kNF * kAuxTile;
    static constexpr int pring = yring + kNF * kCyTile;
    static constexpr int stats = pring + 2 * kPBufBytes;
    static constexpr int c3 = stats + 2 * 4 * 32 * 8;
    static constexpr int total = c3 + 128 * 4;
};
static_assert(RetrLdsHL::pring % 512 == 0 && RetrLdsHL::total <= 160 * 1024, "LDS layout");

#ifdef SVPS_RETR_STAMP
// diagnostic build only (tools/retr_stamps.py): s_memtime stamps of one workgroup's producer 0 and consumer 0, iterations
// 8 .. 15, plus (s_memtime, s_memrealtime) around the loop of every workgroup's wave 0 for the in-kernel clock
__device__ unsigned long long retr_stamps[2][8][8];      // [producer / consumer][iteration - 8][point]
__device__ unsigned long long retr_clock[4096][4];       // [workgroup][memtime0, realtime0, memtime1, realtime1]
#define RETR_STAMP(role, pt)                                                                          \
    do {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        if (blockIdx.x == 3 && blockIdx.y == 2 && sb == 0 && it >= 8 && it < 16 && lane == 0)         \
            retr_stamps[role][it - 8][pt] = __builtin_amdgcn_s_memtime();                             \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    } while (0)
#else
#define RETR_STAMP(role, pt) do {} while (0)
#endif

// ABL: timing-only ablations (env SVPS_RETR_ABLATE), outputs wrong. 1: DMA + barriers only  2: producers only  4: consumers only
//      8: no bf16 -> fp16 conversion
// EXT (more than 128 slots, LP = 256): the launch covers the `L` slots starting at row `slot_off`; the per-pixel softmax
// statistics over ALL slots come from `ext_stats` ([T, HW] of (max logit, 1 / sum of exponentials), written by
// retr_logit_stats_kernel) instead of the exchange between the four producers.
//
// Tile order. A tile is 32 consecutive pixels of one image row; a workgroup walks DOWN a 32-pixel-wide column strip
// (tile id = strip * H + row, a chunk = a run of tile ids), so the Cx terms of its lanes' pixel columns stay in 16 registers for
// a whole strip and the only per-tile position data is the 512-byte Cy row, which arrives by LDS-DMA with the feature
// tile. rstd_k / rstd_v come out of the aux tile (retr_stats.hip stores them there as raw fp32). The producers therefore
// issue NO global loads per tile (they cost 60 - 100 cycles of issue each: ten of them were a third of the tile time).
//
// Schedule (one workgroup barrier B(it) per tile; P(i), stats(i) = results of tile i):
//   producer, iteration it:  MFMA chain of tile it  ||  softmax finish of tile it-1 (reads stats(it-1), writes P(it-1))
//                            then the softmax head of tile it: logits, block maximum, exponentials, block sum -> stats(it)
//   consumer, iteration it:  LDS-DMA of batch it+3;  A += P(it-2) f(it-2) (36 MFMA)
// The exponentials of a tile stay in 16 registers across the barrier; the statistics buffer and the P ring are double-buffered.
// PHL (precision form, L <= 128): P * rstd_v carried as fp16 hi + lo - a second P tile per buffer, the consumers' 18 MFMAs per tile
// twice. Carried as ONE fp16 it is the largest term of the fused retriever's error against float64 (8.7e-4 of 1.1e-3).
// HL (round 4, reference precision on the matrix cores; with PHL): the level map itself is fp16 hi + lo (two planes, f = hi + lo to 22
// bits). A tile is then SIXTEEN pixels: rows 0 .. 15 of the 16-KiB LDS tile are their hi rows, rows 16 .. 31 their lo rows - the ring,
// the swizzle, every fragment address and the barrier schedule stay those of the 32-pixel form (a 32-pixel tile of both planes would
// need twice the ring, which the 160 KiB do not have next to the two P buffers). What changes:
//   producers: the chain yields [Q''.f_hi | Q''.f_lo] in the two column halves of the accumulator; one v_permlane16_swap + add per
//              register folds them (both halves then hold the logits of the same 16 pixels, so the P tile's rows 16 .. 31 repeat rows
//              0 .. 15 - exactly the A operand the consumers' lo k-step needs); Cy + Cx start in the hi half only
//   consumers: k-step 0 = hi rows (P_hi f_hi + P_lo f_hi + the aux block), k-step 1 = lo rows (P_hi f_lo only): 26 MFMAs per tile;
//              waves 0, 1 stage the hi rows, waves 2, 3 the lo rows
// Matrix work per pixel: producers 2 x, consumers 2.9 x the default form - the price of the reference's precision.
template <int ABL = 0, bool EXT = false, bool PHL = false, bool HL = false>
__global__ __launch_bounds__(512) void retr_attn_kernel(
    const __bf16* __restrict__ qh,      // [T, LP, 256]  hi(Q''), rows >= the real slot count zero
    const __bf16* __restrict__ ql,      // [T, LP, 256]  lo(Q'')
    const float* __restrict__ cy,       // [T, H, LP]    Q''[:, 0:128] . ytab[y] + a'
    const float* __restrict__ cx,       // [T, W, LP]    Q''[:, 128:256] . xtab[x]
    const float* __restrict__ c3g,      // [T, LP]  log2(e) * q . beta_k; -1e30 in the padded rows
    const __bf16* __restrict__ feat,    // [T, HW, 256]
    const __bf16* __restrict__ aux,     // [T, HW, 8]    retr_stats.hip: 16-byte rows {1, hi sigma_v, lo sigma_v, 0 (fp16), rstd_k, rstd_v (fp32)}
    float* __restrict__ partial,        // [T, C, Lrow, 260]
    int L, int HW, int H, int W, int tiles_per_chunk, int LP, int Lrow, int slot_off,
    const float2* __restrict__ ext_stats, int map_f16,          // map_f16: the map is fp16 already (no conversion in LDS)
    const __bf16* __restrict__ feat_lo) {                       // HL: the lo plane [T, HW, 256] fp16 (feat is the hi plane)
    static_assert(!HL || PHL, "the hi + lo map form goes with hi + lo probabilities");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    using Lds = std::conditional_t<HL, RetrLdsHL, RetrLdsT<PHL ? 2 : 1>>;
    constexpr int A = Lds::kA;
    constexpr int NF = Lds::kNF;                                // ring depth (feature / aux / Cy tiles)
    constexpr int TPX = HL ? 16 : kTilePx;                      // pixels per tile
    constexpr int kPBuf = Lds::kPBufBytes;                      // one P buffer: the hi tile (and the lo tile behind it / in its blocks' upper halves)
    constexpr int kPLo = Lds::kPLo;

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sb = w & 3;
    const bool consumer = w >= 4;
    const int r = lane & 31, h = lane >> 5;
    const int rp = HL ? (r & 15) : r;                           // this lane's pixel inside the tile
    const int C = gridDim.x;
    int t = blockIdx.y, c = blockIdx.x;
    if ((gridDim.y & 7) == 0) {
        // XCD-aware frame placement (speed only, any mapping is correct): workgroups are handed to the 8 XCDs round-robin
        // by linear id; here all chunks of a frame go to ONE XCD, so an L2 holds the position tables and the slot operands
        // of the two or three frames its 32 CUs work on instead of those of every frame in flight.
        const int b = blockIdx.y * C + blockIdx.x;
        const int n = b >> 3;
        t = (b & 7) + 8 * (n / C);
        c = n % C;
    }
    const int tiles = ((W + TPX - 1) / TPX) * H;
    const int tid0 = c * tiles_per_chunk;
    int nt = tiles - tid0;
    nt = nt < tiles_per_chunk ? nt : tiles_per_chunk;           // >= 1 by construction of the grid
    const int strip0 = tid0 / H, row0 = tid0 - strip0 * H;      // first tile of the chunk
    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>((SVPS_LDS const void*)smem);

    float* c3l = reinterpret_cast<float*>(smem + Lds::c3);
    if (threadIdx.x < 128) c3l[threadIdx.x] = c3g[(size_t)t * LP + slot_off + threadIdx.x];
#ifdef SVPS_RETR_STAMP
    const int wg_lin = blockIdx.y * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0 && wg_lin < 4096) {
        retr_clock[wg_lin][0] = __builtin_amdgcn_s_memtime();
        retr_clock[wg_lin][1] = __builtin_amdgcn_s_memrealtime();
    }
#endif

    if (!consumer) {
        // ================================ producer =============================================
        f16x8 qfh[16], qfl[16];
        {
            const size_t row = ((size_t)t * LP + slot_off + 32 * sb + r) * kD + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                qfh[ks] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(qh + row + 16 * ks));
                qfl[ks] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(ql + row + 16 * ks));
            }
        }
        const int slot0 = 32 * sb + 4 * h;                          // accumulator register 4 g + j <-> slot row slot0 + 8 g + j
        const int key = (r >> 1) & 3;
        // HL (round 5): after the chain lane r < 16 holds [Q''.f_hi + Cy + Cx] and lane r + 16 [Q''.f_lo] of the SAME pixel for all 16 slot
        // registers. The fold EXCHANGES register halves instead of duplicating sums (v_permlane16_swap(s[i], s[i + 8])): lane r < 16 ends up
        // with the full logits of registers 0 .. 7 (slot groups g = 0, 1), lane r + 16 with those of registers 8 .. 15 (g = 2, 3) - the
        // softmax head and the finish then run on EIGHT registers per lane instead of sixteen duplicated ones (the producers are the
        // critical path of this form: dropping consumer MFMAs changed nothing, halving this vector work does).
        constexpr int NG = HL ? 2 : 4;                              // slot groups of four per lane behind the chain
        constexpr int NE = 4 * NG;
        const int gsel = HL ? 2 * ((r >> 4) & 1) : 0;               // first slot group of this lane: g = gsel + g'
        const uint32_t lane_row = lds0 + Lds::fring + r * kRowBytes + ((h ^ swz(r)) << 4);
        // tables through buffer descriptors (scalar registers) + 32-bit lane offsets: no 64-bit pointers in vector registers
        auto uniform_rsrc = [](const void* p, int bytes) {          // every word provably wave-uniform: no waterfall loops
            const uint64_t a = reinterpret_cast<uint64_t>(p);
            // readfirstlane returns a SIGNED int: widen through uint32_t or the low word sign-extends into the high one
            const uint64_t u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32)) << 32) |
                               (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a);
            return __builtin_amdgcn_make_buffer_rsrc((void*)u, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
        };
        const __amdgpu_buffer_rsrc_t cyr = uniform_rsrc(cy + (size_t)t * H * LP + slot_off, H * LP * 4 - slot_off * 4);
        const __amdgpu_buffer_rsrc_t cxr = uniform_rsrc(cx + (size_t)t * W * LP + slot_off, W * LP * 4 - slot_off * 4);
        const __amdgpu_buffer_rsrc_t exr = uniform_rsrc(EXT ? (const void*)(ext_stats + (size_t)t * HW) : (const void*)cy, EXT ? HW * 8 : 0);

        // Cx of this lane's pixel column: constant down a strip (clamped past the right edge: those pixels are masked)
        f32x4 cxv[4];
        auto load_cx = [&](int strip) {
            int xx = TPX * strip + rp;
            xx = xx < W ? xx : W - 1;
            const int xo = (xx * LP + slot0) * 4;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                cxv[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(cxr, xo + 32 * g, 0, 0));
                if (HL && (r & 16)) cxv[g] = f32x4{0.f, 0.f, 0.f, 0.f};       // the lo half of the accumulator starts from zero
            }
        };
        const float cy_on = (HL && (r & 16)) ? 0.f : 1.f;
        int ts = strip0, ty = row0;                                 // strip / image row of tile `it`
        f32x16 cinit;                                               // Cy + Cx of tile `it`: the initial value of its accumulator
        f32x2 ext_n = {0.f, 0.f};                                   // EXT: statistics of this lane's pixel, requested one tile ahead
        auto request_ext = [&](int strip, int row) {
            if constexpr (EXT) {
                int px = row * W + TPX * strip + rp;
                px = px < HW ? px : HW - 1;
                ext_n = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(exr, px * 8, 0, 0));
            }
        };
        load_cx(ts);
        {   // tile 0: its Cy row straight from global memory (the rows of the later tiles arrive by LDS-DMA, one batch early)
            const int yo = (ty * LP + slot0) * 4;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 cyv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(cyr, yo + 32 * g, 0, 0));
#pragma unroll
                for (int j = 0; j < 4; ++j) cinit[4 * g + j] = cyv[j] * cy_on + cxv[g][j];
            }
        }
        request_ext(ts, ty);

        f32x16 e;                                                   // exponentials of the previous tile (relative to its block maximum)
        float mloc_p = 0.f, tau_p = 0.f;
        bool live_p = false;
        f32x2 ext_p = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 16; ++i) e[i] = 0.f;
        float2* stats = reinterpret_cast<float2*>(smem + Lds::stats);

        // first fragment group and (rstd_k, rstd_v) of a tile: requested at the END of the previous iteration (the consumers
        // let a barrier pass only when the batch after the current one has landed), so their latency runs under the barrier
        f16x8 kf0[4];
        f32x2 rt = {0.f, 0.f};
        constexpr int kOrd[4] = {0, 8, 1, 9};                       // group g: k-steps 2g, 2g + 8, 2g + 1, 2g + 9
        // LDS byte address of this lane's 16-B chunk of k-step ks = 8 a + b in a tile: (tile + lane_row) ^ (b << 5), + 256 a
        auto frag = [&](uint32_t tb, int ks) {
            return *reinterpret_cast<SVPS_LDS const f16x8*>((uintptr_t)((tb ^ ((ks & 7) << 5)) + 256 * (ks >> 3)));
        };
        auto prefetch = [&](int tile) {
            const uint32_t slot = (uint32_t)(tile % NF);
            const uint32_t tb = lane_row + slot * kTileBytes;
#pragma unroll
            for (int u = 0; u < 4; ++u) kf0[u] = frag(tb, kOrd[u]);
            // (rstd_k, rstd_v) of this lane's pixel: bytes 8 .. 15 of its aux row
            rt = *reinterpret_cast<SVPS_LDS const f32x2*>((uintptr_t)(lds0 + Lds::aring + slot * kAuxTile + r * kAuxRow + 8));
        };

        // One iteration. CHAIN: tile `it` exists (MFMA chain + softmax head); P2: tile it-1 exists (softmax finish).
        auto body = [&](int it, auto chain_tag, auto p2_tag) {
            constexpr bool CHAIN = decltype(chain_tag)::value, P2 = decltype(p2_tag)::value;
            const uint32_t tb = lane_row + (uint32_t)(it % NF) * kTileBytes;
            // ---- softmax finish of tile it-1, part 1: the four blocks' statistics (requested now, used under the second MFMA group)
            float fac = 0.f;
            float2 st_w[4];
            if constexpr (P2 && !EXT) {
                const float2* st = stats + ((it - 1) & 1) * 128 + r;
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) st_w[ww] = st[ww * 32];
            }
            auto p2_factor = [&]() {                                // normalisation factor of this lane's pixel: P * rstd_v = e * fac
                if constexpr (EXT) {
                    fac = ext_p[1] * tau_p;
                } else {
                    float mall = kNegBig;
#pragma unroll
                    for (int ww = 0; ww < 4; ++ww) mall = fmaxf(mall, st_w[ww].x);
                    float den = 0.f;
#pragma unroll
                    for (int ww = 0; ww < 4; ++ww) den += st_w[ww].y * __builtin_amdgcn_exp2f(st_w[ww].x - mall);
                    fac = __builtin_amdgcn_exp2f(mloc_p - mall) * __builtin_amdgcn_rcpf(den) * tau_p;
                }
                if (!live_p) fac = 0.f;                             // pixels past the right edge of the map
            };
            char* prow = smem + Lds::pring + ((it - 1) & 1) * kPBuf + sb * 2048 + r * 64 + 8 * h;
            char* prowh = smem + Lds::pring + ((it - 1) & 1) * kPBuf + sb * 2048 + (r & 15) * 64 + 8 * h;   // HL: row of this lane's pixel
            auto p2_store = [&](int g) {                            // four slots of P(it-1) = e * fac, fp16 (HL: g = 0, 1 of this lane's groups)
                f16x4 ph;
                if constexpr (PHL) {
                    f16x4 pl;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float x = e[4 * g + j] * fac;
                        asm volatile("" : "+v"(x));          // ONE fp32 value for both halves (hipcc otherwise rounds hi from the fp32
                                                             // product and lo from the exact product minus its own hi: v_fma_mixlo_f16)
                        ph[j] = (_Float16)x;
                        pl[j] = (_Float16)(x - (float)ph[j]);
                    }
                    if constexpr (HL) {
                        // this lane's slot group gsel + g of pixel r & 15: the hi row of that pixel, the lo row 1 KiB behind it (round 6: sixteen
                        // rows per slot block; the consumers' lo k-step re-uses the A fragment of k-step 0)
                        char* pr = prowh + (((gsel + g) ^ key) * 16);
                        *reinterpret_cast<f16x4*>(pr) = ph;
                        *reinterpret_cast<f16x4*>(pr + kPLo) = pl;
                        return;
                    }
                    *reinterpret_cast<f16x4*>(prow + kPLo + ((g ^ key) * 16)) = pl;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) ph[j] = (_Float16)(e[4 * g + j] * fac);
                }
                *reinterpret_cast<f16x4*>(prow + ((g ^ key) * 16)) = ph;
            };
            f32x16 s = cinit;
            f32x4 c3v[4];
            const float rk_c = rt[0] * kLog2e, tau_c = rt[1] * kPScale;      // common.h: the probabilities carry 2^7
            if constexpr (CHAIN) {
                // row fragments in groups of four, double-buffered: the reads of group g + 1 (after the last group: the c3 terms)
                // and a part of the softmax finish of tile it-1 run in the shadow of the 8 MFMAs of group g
                f16x8 kf[2][4];
#pragma unroll
                for (int u = 0; u < 4; ++u) kf[0][u] = kf0[u];
#pragma unroll
                for (int grp = 0; grp < 4; ++grp) {
                    if (grp < 3) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) kf[(grp + 1) & 1][u] = frag(tb, 2 * (grp + 1) + kOrd[u]);
                    } else {
#pragma unroll
                        for (int g = 0; g < NG; ++g) c3v[g] = *reinterpret_cast<const f32x4*>(c3l + slot0 + 8 * (gsel + g));
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qfh[2 * grp + kOrd[u]], kf[grp & 1][u], s, 0, 0, 0);
                        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qfl[2 * grp + kOrd[u]], kf[grp & 1][u], s, 0, 0, 0);
                    }
                    if constexpr (P2) {
                        if (grp == 0) p2_factor();
                        if constexpr (HL) {
                            if (grp == 1) p2_store(0);
                            if (grp == 2) p2_store(1);
                        } else {
                            if (grp == 1) { p2_store(0); p2_store(1); }
                            if (grp == 2) { p2_store(2); p2_store(3); }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {                   // one MFMA, then its share of the other work
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (u < 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        // group 0: its vector work (the normalisation factor) hangs on the statistics read issued at the top of
                        // the iteration - behind the first four MFMAs, so that an in-order wave does not park the chain on it
                        // (placing it a group later, with the stores in groups 2 and 3, measures the same)
                        if (grp == 0) { if (u >= 4) __builtin_amdgcn_sched_group_barrier(0x002, 8, 0); }
                        else __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if constexpr (P2) {
                p2_factor();
#pragma unroll
                for (int g = 0; g < NG; ++g) p2_store(g);
            }
            RETR_STAMP(0, 2);
            if constexpr (!CHAIN) return;
            if constexpr (HL) {
                // columns r < 16: Q''.f_hi + Cy + Cx of pixel r; columns r >= 16: Q''.f_lo of pixel r - 16. v_permlane16_swap exchanges the
                // odd 16-lane rows of its first operand with the even rows of its second: swap(s[i], s[i + 8]) returns
                // [s_i.row0, s_{i+8}.row0, s_i.row2, s_{i+8}.row2] and [s_i.row1, s_{i+8}.row1, s_i.row3, s_{i+8}.row3], whose sum is the
                // full logit of register i in the even rows (lanes r < 16) and of register i + 8 in the odd rows (lanes r >= 16)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(s[i]), __float_as_uint(s[i + 8]), false, false);
                    s[i] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
                }
            }
            // ---- softmax head of tile it
            // log2(e) * S = (log2(e) rstd_k) * (Q''.f + Cy + Cx) + c3'. Rows past the real slot count need no masking: their
            // Q'', Cy, Cx are zero and their c3' is -1e30 (retr_query_prep), so they come out as -1e30 and exp2 to exactly 0.
            live_p = TPX * ts + rp < W;
            ++ty;
            if (ty == H) { ty = 0; ++ts; if (it + 1 < nt) load_cx(ts); }
            const bool more = it + 1 < nt;
            float mloc = kNegBig;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s[4 * g + j] = fmaf(rk_c, s[4 * g + j], c3v[g][j]);
                    if constexpr (!EXT) mloc = fmaxf(mloc, s[4 * g + j]);
                }
            }
            if constexpr (HL && !EXT) {                             // the two lanes of a pixel hold different slot groups: combine them
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(mloc), __float_as_uint(mloc), false, false);
                mloc = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            // Cy row of tile it+1 (staged with batch `it`): requested now, added to Cx after the exponentials
            f32x4 cyv[4];
            const float* cyl = reinterpret_cast<const float*>(smem + Lds::yring + ((it + 1) % NF) * kCyTile) + slot_off + slot0;
#pragma unroll
            for (int g = 0; g < 4; ++g) cyv[g] = *reinterpret_cast<const f32x4*>(cyl + 8 * g);
            const f32x2 ext_c = ext_n;
            if constexpr (EXT) mloc = ext_c[0];                     // statistics over all slots are known (log2 domain as well)
            else mloc = ra_half_swap_max(mloc);
            if (more) request_ext(ts, ty);
            RETR_STAMP(0, 6);
#pragma unroll
            for (int i = 0; i < NE; ++i) s[i] -= mloc;
#pragma unroll
            for (int i = 0; i < NE; ++i) s[i] = __builtin_amdgcn_exp2f(s[i]);
            if constexpr (!EXT) {
                float sl[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) sl[i] = HL ? s[i] + s[4 + i] : (s[i] + s[4 + i]) + (s[8 + i] + s[12 + i]);
                float sloc = (sl[0] + sl[1]) + (sl[2] + sl[3]);
                if constexpr (HL) {
                    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(sloc), __float_as_uint(sloc), false, false);
                    sloc = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
                }
                sloc = ra_half_swap_sum(sloc);
                if (h == 0) stats[(it & 1) * 128 + sb * 32 + r] = make_float2(mloc, sloc);
            }
            e = s;
            mloc_p = mloc;
            tau_p = tau_c;
            ext_p = ext_c;
            RETR_STAMP(0, 3);
            // ---- next tile: position terms, first fragment group
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j) cinit[4 * g + j] = cyv[g][j] * cy_on + cxv[g][j];
            if (more) prefetch(it + 1);
            RETR_STAMP(0, 5);
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        constexpr bool kRun = ABL != 1 && ABL != 4;
        wg_barrier();                                               // B(0)
        if constexpr (kRun) { prefetch(0); body(0, T_{}, F_{}); }
        for (int it = 1; it < nt; ++it) {
            RETR_STAMP(0, 0);
            wg_barrier();                                           // B(it)
            RETR_STAMP(0, 1);
            if constexpr (kRun) body(it, T_{}, T_{});
        }
        wg_barrier();                                               // B(nt)
        if constexpr (kRun) body(nt, F_{}, T_{});
        wg_barrier();                                               // B(nt + 1)
#ifdef SVPS_RETR_STAMP
        if (threadIdx.x == 0 && wg_lin < 4096) {
            retr_clock[wg_lin][2] = __builtin_amdgcn_s_memtime();
            retr_clock[wg_lin][3] = __builtin_amdgcn_s_memrealtime();
        }
#endif
        return;
    }

    // =================================== consumer ===============================================
    // HL: tile rows 0 .. 15 come from the hi plane (waves 0, 1), rows 16 .. 31 from the lo plane (waves 2, 3), same 16 pixels
    const u32x4 frs = ra_make_srd((HL && sb >= 2 ? feat_lo : feat) + (size_t)t * HW * kD, (uint32_t)HW * kRowBytes);
    const u32x4 ars = ra_make_srd(aux + (size_t)t * HW * 8, (uint32_t)HW * kAuxRow);
    const u32x4 yrs = ra_make_srd(cy + (size_t)t * H * LP, (uint32_t)(H * LP) * 4u);
    int voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * sb + 2 * i + h;                     // row of the LDS tile
        voff[i] = (HL ? (row & 15) : row) * kRowBytes + (((lane & 31) ^ swz(row)) * 16);
    }
    const int nb = 4 + ((sb == 0 || sb == 2) ? 1 : 0);          // DMA instructions of one batch of this wave
    int ds = strip0, dy = row0;                                 // strip / image row of the next batch
    auto issue_batch = [&](int b) {
        if (b >= nt) return;
        const uint32_t st = __builtin_amdgcn_readfirstlane(lds0 + Lds::fring + (b % NF) * kTileBytes + sb * 4 * 1024);
        const int px0 = dy * W + TPX * ds;
        const int soff = __builtin_amdgcn_readfirstlane(px0 * kRowBytes);
        if (px0 + TPX <= HW) {
#pragma unroll
            for (int i = 0; i < 4; ++i) ra_dma16(frs, st + i * 1024, voff[i], soff);
        } else {                                                 // last row of a ragged strip: clamp the source rows (their P is 0)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 8 * sb + 2 * i + h;
                const int prow_ = HL ? (row & 15) : row;
                const int src = px0 + prow_ < HW ? prow_ : HW - 1 - px0;
                ra_dma16(frs, st + i * 1024, src * kRowBytes + (((lane & 31) ^ swz(row)) * 16), soff);
            }
        }
        ++dy;
        if (dy == H) { dy = 0; ++ds; }
        if (sb == 0) {                                           // aux tile: 32 rows of 16 B (lanes >= 32 repeat them); rows past the frame read zeros
            // HL: 16 pixels, rows 16 .. 31 repeat rows 0 .. 15 (the producers' lanes r and r + 16 belong to the same pixel)
            const uint32_t sa = __builtin_amdgcn_readfirstlane(lds0 + Lds::aring + (b % NF) * kAuxTile);
            ra_dma16(ars, sa, (px0 + (lane & (TPX - 1))) * kAuxRow, 0);
        } else if (sb == 2) {                                    // Cy row of tile b + 1 (1 KiB from the start of its image row)
            const uint32_t sy = __builtin_amdgcn_readfirstlane(lds0 + Lds::yring + ((b + 1) % NF) * kCyTile);
            ra_dma16_cached(yrs, sy, dy * LP * 4 + lane * 16);
        }
    };
#pragma unroll
    for (int b = 0; b < A; ++b) issue_batch(b);

    f32x16 o[8], oa;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        oa[i] = 0.f;
#pragma unroll
        for (int db = 0; db < 8; ++db) o[db][i] = 0.f;
    }
    // Fragment addresses (LDS byte addresses): everything that depends on the k-step, the channel block or hi / lo is an
    // instruction offset or one XOR; per tile the ring slot is added to five per-lane terms.
    //   value  V[pixels 16 ks + 8 h' .. + 8][channel 32 db + n]: rows rowl (+4), chunk (4 db + cl) ^ swz(row)
    //          = vt + 8192 ks + 256 (db >> 2) + (lane_v{0,1} ^ ((db & 3) << 6))
    //   P      rows rowl (+4) of 64 B, chunk cl ^ ((row >> 1) & 3): pt + 1024 ks + lane_p{0,1}
    //   aux    rows rowl (+4) of 64 B, linear: at + 1024 ks (+ 256) + lane_a
    const int g2 = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    const int cl = 2 * (g2 & 1) + (pp >> 1), sub = 8 * (pp & 1), rowl = 8 * (g2 >> 1) + qq;
    const uint32_t lane_v0 = rowl * kRowBytes + (((cl ^ (2 * (g2 >> 1))) + 4 * qq) << 4) + sub;
    const uint32_t lane_v1 = (rowl + 4) * kRowBytes + (((cl ^ (2 * (g2 >> 1) + 1)) + 4 * qq) << 4) + sub;
    const uint32_t lane_p0 = sb * 2048 + sub + rowl * 64 + ((cl ^ (qq >> 1)) << 4);
    const uint32_t lane_p1 = sb * 2048 + sub + (rowl + 4) * 64 + ((cl ^ (qq >> 1) ^ 2) << 4);
    // aux block: the row's FP16 words 0 .. 3 are columns 0 .. 3 of the ninth channel block; the lanes that feed the other 28
    // columns read words 4 .. 7 (the two fp32 statistics as bit patterns) or repeat words 0 .. 3: those columns only reach
    // accumulator columns nobody stores
    const uint32_t lane_a = rowl * kAuxRow + ((cl == 0 && sub != 0) ? 8 : 0);
    auto tr = [](uint32_t a) {
        return __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16(reinterpret_cast<SVPS_LDS fp16x4_gcc*>((uintptr_t)a)));
    };
    auto cat = [](f16x4 a, f16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); };
    // A += P f for one tile: 4 steps (k-step, half of the channel blocks) of 4 MFMA (+ 1 for the aux block), the fragments of
    // step q + 1 requested before the MFMAs of step q; those of step 0 before the LDS-DMA of the iteration is issued
    f16x8 ah[2], al[2], af[2], vf[2][4];
    uint32_t p0 = 0, p1 = 0, v0 = 0, v1 = 0, aa = 0;
    auto vfrag = [&](int ks, int db) {
        const uint32_t o_ = 8192 * ks + 256 * (db >> 2);
        return cat(tr((v0 ^ ((db & 3) << 6)) + o_), tr((v1 ^ ((db & 3) << 6)) + o_));
    };
    auto pv_begin = [&](int j) {
        const uint32_t pt = lds0 + Lds::pring + (j & 1) * kPBuf, vt = lds0 + Lds::fring + (j % NF) * kTileBytes;
        const uint32_t at = lds0 + Lds::aring + (j % NF) * kAuxTile;
        p0 = pt + lane_p0, p1 = pt + lane_p1, v0 = vt + lane_v0, v1 = vt + lane_v1, aa = at + lane_a;
        ah[0] = cat(tr(p0), tr(p1));
        if constexpr (PHL) al[0] = cat(tr(p0 + kPLo), tr(p1 + kPLo));
        af[0] = cat(tr(aa), tr(aa + 4 * kAuxRow));
#pragma unroll
        for (int u = 0; u < 4; ++u) vf[0][u] = vfrag(0, u);
    };
    auto pv_rest = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ks = q >> 1, half = q & 1;
            if (q < 3) {
#pragma unroll
                for (int u = 0; u < 4; ++u) vf[(q + 1) & 1][u] = vfrag((q + 1) >> 1, 4 * ((q + 1) & 1) + u);
            }
            if (q == 1) {
                if constexpr (HL) ah[1] = ah[0];                 // the lo rows belong to the SAME sixteen pixels: the same probabilities
                else ah[1] = cat(tr(p0 + 1024), tr(p1 + 1024));
                if constexpr (PHL && !HL) al[1] = cat(tr(p0 + kPLo + 1024), tr(p1 + kPLo + 1024));
                if constexpr (!HL) af[1] = cat(tr(aa + 16 * kAuxRow), tr(aa + 20 * kAuxRow));
            }
            __builtin_amdgcn_sched_barrier(0);
            // HL: k-step 1 holds the lo rows of the SAME pixels: P_hi f_lo only (P_lo f_lo is below fp32 resolution), no aux block
            const bool lo_step = HL && ks == 1;                      // (compile-time after unrolling)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                o[4 * half + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], vf[q & 1][u], o[4 * half + u], 0, 0, 0);
                if (PHL && !lo_step) o[4 * half + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], vf[q & 1][u], o[4 * half + u], 0, 0, 0);
            }
            if (half == 0 && !lo_step) {
                oa = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], af[ks], oa, 0, 0, 0);
                if (PHL) oa = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], af[ks], oa, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // This wave's four pieces of feature tile b, bf16 -> fp16 in place (exact: bf16 values are fp16 values for |f| in
    // [6.1e-5, 65504]; smaller ones lose bits they cannot influence anything with, larger ones saturate). Every matrix
    // instruction of the kernel then runs fp16 x fp16: Q'' as fp16 hi + lo carries 22 bits, P * rstd_v ONE fp16 (11 bits,
    // rounding errors average out over the pixel sum) instead of bf16 hi + lo: half the MFMAs on the value side.
    auto convert_batch = [&](int b) {
        if (b >= nt || ABL == 8 || map_f16) return;
        const uint32_t st = lds0 + Lds::fring + (b % NF) * kTileBytes + sb * 4096 + lane * 16;
        u32x4 w_[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w_[i] = *reinterpret_cast<SVPS_LDS const u32x4*>((uintptr_t)(st + i * 1024));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const fp16x2_t pk = __builtin_amdgcn_cvt_pkrtz(__uint_as_float(w_[i][k] << 16), __uint_as_float(w_[i][k] & 0xffff0000u));
                w_[i][k] = __builtin_bit_cast(uint32_t, pk);
            }
            *reinterpret_cast<SVPS_LDS u32x4*>((uintptr_t)(st + i * 1024)) = w_[i];
        }
    };

    for (int it = 0; it <= nt + 1; ++it) {
        // batches <= it+1 (feature + aux tiles, the Cy row of the tile after) landed for this wave: all but the youngest batch
        // (the producers request the first fragments of tile it+1 at the end of iteration it)
        RETR_STAMP(1, 0);
        if (it + A - 1 < nt) wait_vm_dyn(nb * (A - 2));
        else wait_vm<0>();
        if (it == 0) convert_batch(0);
        convert_batch(it + 1);
        RETR_STAMP(1, 1);
        wg_barrier();                                            // B(it)
        RETR_STAMP(1, 2);
        const bool work = it >= 2 && ABL != 1 && ABL != 2;
        if (work) pv_begin(it - 2);
        issue_batch(it + A);
        RETR_STAMP(1, 3);
        if (work) pv_rest();
        RETR_STAMP(1, 5);
    }

    float* dst = partial + (((size_t)t * C + c) * Lrow + slot_off) * kPartRow;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int slot = 32 * sb + acc_row(i, h);
        if (slot < L) {
#pragma unroll
            for (int db = 0; db < 8; ++db) dst[(size_t)slot * kPartRow + 32 * db + r] = o[db][i];
            if (r < 4) dst[(size_t)slot * kPartRow + 256 + r] = oa[i];     // aux columns 0 .. 2 (column 3 is zero); the rest is not data
        }
    }
}

// Per-pixel softmax statistics over up to 256 slots (first kernel of the path for more than 128 slots):
// stats[t, p] = (max_l S[l, p], 1 / sum_l exp(S[l, p] - max)) with S exactly as in retr_attn_kernel (same fp16 operands, same
// strip tile order, same position-term handling). Eight waves = eight slot blocks, each with its Q'' hi / lo block resident.
// Every wave stages two 1-KiB pieces of every feature tile (LDS-DMA, 4 tiles ahead in a 6-deep ring) and converts them to
// fp16 when they land; wave 7 also stages the Cy row of the tile (1 KiB = 256 slots), wave 6 the 32 rstd_k values. ONE
// barrier per tile: the statistics of the eight blocks are double-buffered and combined at the START of the next
// iteration (wave w: pixels 4w .. 4w+3). Reads the map once, writes 8 B per pixel. No global load in the main loop.
struct LStatsLds {
    static constexpr int kA = 4;                                // tiles ahead
    static constexpr int kNF = kA + 2;                          // ring depth: tile it (compute), it+1 (converted), it+2 .. it+A+1 in flight
    static constexpr int ring = 0;
    static constexpr int yring = kNF * kTileBytes;              // kNF x 1 KiB Cy rows
    static constexpr int kring = yring + kNF * 1024;            // kNF x 256 B rstd_k of the tile's pixels (64 lanes x 4 B, lanes >= 32 repeat)
    static constexpr int stats = kring + kNF * 256;             // [2][8][32] float2
    static constexpr int total = stats + 2 * 8 * 32 * 8;
};

// HL (round 5, reference precision for more than 128 slots): the map as fp16 hi + lo planes in the 16-pixel tile of retr_attn_kernel<.., HL>
// (tile rows 0 .. 15 = hi rows, staged by waves 0 .. 3 from `feat`; rows 16 .. 31 = lo rows of the same pixels, waves 4 .. 7 from `feat_lo`):
// the chain yields [Q''.f_hi | Q''.f_lo] in the two column halves, folded with one v_permlane16_swap + add per register (Cy + Cx start in
// the hi half only); wave w combines pixels 2w, 2w + 1.
template <bool HL>
__global__ __launch_bounds__(512) void retr_logit_stats_kernel(
    const _Float16* __restrict__ qh, const _Float16* __restrict__ ql,  // [T, 256, 256]
    const float* __restrict__ cy, const float* __restrict__ cx,        // [T, H, 256], [T, W, 256]
    const float* __restrict__ c3g,                                     // [T, 256]
    const __bf16* __restrict__ feat, const __bf16* __restrict__ aux,   // aux: the 16-byte rows of retr_stats.hip (rstd_k = bytes 8 .. 11)
    float2* __restrict__ out,                                          // [T, HW]
    int L, int HW, int H, int W, int tiles_per_chunk, int map_f16,
    const __bf16* __restrict__ feat_lo) {                              // HL: the lo plane (feat is the hi plane)
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    using Lds = LStatsLds;
    constexpr int NF = Lds::kNF, A = Lds::kA, LP = 256;
    constexpr int TPX = HL ? 16 : kTilePx;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int rp = HL ? (r & 15) : r;                                  // this lane's pixel inside the tile
    const int t = blockIdx.y, c = blockIdx.x;
    const int tiles = ((W + TPX - 1) / TPX) * H;
    const int tid0 = c * tiles_per_chunk;
    int nt = tiles - tid0;
    nt = nt < tiles_per_chunk ? nt : tiles_per_chunk;
    const int strip0 = tid0 / H, row0 = tid0 - strip0 * H;
    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>((SVPS_LDS const void*)smem);

    f16x8 qfh[16], qfl[16];
    {
        const size_t row = ((size_t)t * LP + 32 * w + r) * kD + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            qfh[ks] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(qh + row + 16 * ks));
            qfl[ks] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(ql + row + 16 * ks));
        }
    }
    const int slot0 = 32 * w + 4 * h;
    f32x4 c3v[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) c3v[g] = *reinterpret_cast<const f32x4*>(c3g + (size_t)t * LP + slot0 + 8 * g);
    // Make hipcc consume (and therefore wait for) every register loaded above HERE: its wait-count pass does not see the asm waits
    // and otherwise places `s_waitcnt vmcnt(n)` before the first use of each fragment INSIDE the main loop, where a small n
    // also waits for the LDS-DMA ring.
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) asm volatile("" : "+v"(qfh[ks]), "+v"(qfl[ks]));
#pragma unroll
    for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(c3v[g]));
    wait_vm<0>();

    const u32x4 cys = ra_make_srd(cy + (size_t)t * H * LP, (uint32_t)(H * LP) * 4u);
    const u32x4 cxs = ra_make_srd(cx + (size_t)t * W * LP, (uint32_t)(W * LP) * 4u);
    const u32x4 frs = ra_make_srd((HL && w >= 4 ? feat_lo : feat) + (size_t)t * HW * kD, (uint32_t)HW * kRowBytes);
    const u32x4 krs = ra_make_srd(aux + (size_t)t * HW * 8, (uint32_t)HW * kAuxRow);
    const u32x4 ors = ra_make_srd(out + (size_t)t * HW, (uint32_t)HW * 8u);
    auto ld16 = [](u32x4 srd, int off) {                            // asm + its own wait (no compiler-visible load in the loop)
        f32x4 v;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(off), "s"(srd) : "memory");
        return v;
    };
    f32x4 cxv[4];
    auto load_cx = [&](int strip) {
        int xx = TPX * strip + rp;
        xx = xx < W ? xx : W - 1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            cxv[g] = ld16(cxs, (xx * LP + slot0 + 8 * g) * 4);
            if (HL && (r & 16)) cxv[g] = f32x4{0.f, 0.f, 0.f, 0.f};   // the lo half of the accumulator starts from zero
        }
    };
    const float cy_on = (HL && (r & 16)) ? 0.f : 1.f;
    load_cx(strip0);

    // ---- staging: rows 4w .. 4w+3 of every tile (two pieces); wave 7: the Cy row; wave 6: rstd_k of the 32 pixels
    int voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 4 * w + 2 * i + h;                           // row of the LDS tile (HL: rows 16 .. 31 = lo rows of pixels 0 .. 15)
        voff[i] = (HL ? (row & 15) : row) * kRowBytes + (((lane & 31) ^ swz(row)) * 16);
    }
    const int nb = 2 + (w >= 6 ? 1 : 0);                            // DMA instructions of one batch of this wave
    int ds = strip0, dy = row0;
    auto stage = [&](int tile) {
        if (tile >= nt) return;
        const uint32_t st = __builtin_amdgcn_readfirstlane(lds0 + Lds::ring + (tile % NF) * kTileBytes + w * 2048);
        const int px0 = dy * W + TPX * ds;
        const int soff = __builtin_amdgcn_readfirstlane(px0 * kRowBytes);
        if (px0 + TPX <= HW) {
            ra_dma16(frs, st, voff[0], soff);
            ra_dma16(frs, st + 1024, voff[1], soff);
        } else {                                                     // last row of a ragged strip: clamp the source rows (not stored)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = 4 * w + 2 * i + h;
                const int prow_ = HL ? (row & 15) : row;
                const int src = px0 + prow_ < HW ? prow_ : HW - 1 - px0;
                ra_dma16(frs, st + i * 1024, src * kRowBytes + (((lane & 31) ^ swz(row)) * 16), soff);
            }
        }
        if (w == 7) {
            ra_dma16_cached(cys, __builtin_amdgcn_readfirstlane(lds0 + Lds::yring + (tile % NF) * 1024), dy * LP * 4 + lane * 16);
        } else if (w == 6) {                                         // 4 B per lane: pixels px0 + (lane & 31), clamped into the frame
            int px = px0 + (lane & (TPX - 1));
            px = px < HW ? px : HW - 1;
            uint32_t keep;
            const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + Lds::kring + (tile % NF) * 256);
            asm volatile(
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b32 m0, %1\n\t"
                "s_nop 0\n\t"
                "buffer_load_dword %2, %3, 0 offen lds\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(keep)
                : "s"(dst), "v"(px * kAuxRow + 8), "s"(krs)
                : "memory");
        }
        ++dy;
        if (dy == H) { dy = 0; ++ds; }
    };
    auto convert = [&](int tile) {
        if (tile >= nt || map_f16 || HL) return;
        const uint32_t st = lds0 + Lds::ring + (tile % NF) * kTileBytes + w * 2048 + lane * 16;
        u32x4 w_[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) w_[i] = *reinterpret_cast<SVPS_LDS const u32x4*>((uintptr_t)(st + i * 1024));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                w_[i][k] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(__uint_as_float(w_[i][k] << 16), __uint_as_float(w_[i][k] & 0xffff0000u)));
            *reinterpret_cast<SVPS_LDS u32x4*>((uintptr_t)(st + i * 1024)) = w_[i];
        }
    };
    float2* stats = reinterpret_cast<float2*>(smem + Lds::stats);
    // statistics of tile `tile` (strip fs, row fy): wave w combines the eight blocks for pixels 4w .. 4w+3 and stores them. The
    // store is ALWAYS issued (out-of-range offset when there is nothing to store): the counted vmcnt waits rely on it.
    int fs = strip0, fy = row0;
    auto combine = [&](int tile) {
        const bool have = tile >= 0;
        float mall = kNegBig;
        float2 st_w[8];
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) {
            st_w[ww] = stats[((tile & 1) * 8 + ww) * 32 + r];
            mall = fmaxf(mall, st_w[ww].x);
        }
        float den = 0.f;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) den += st_w[ww].y * __builtin_amdgcn_exp2f(st_w[ww].x - mall);
        const int xx = TPX * fs + rp;
        const int pxs = fy * W + xx;
        const bool mine = have && h == 0 && (HL ? (r < 16 && (r >> 1) == w) : (r >> 2) == w) && xx < W;
        const f32x2 val = {mall, 1.f / den};
        const int so = mine ? pxs * 8 : 0x7ffffff0;                  // out of range -> dropped by the hardware range check
        asm volatile("buffer_store_dwordx2 %0, %1, %2, 0 offen" : : "v"(val), "v"(so), "s"(ors) : "memory");
        if (have) {
            ++fy;
            if (fy == H) { fy = 0; ++fs; }
        }
    };

    // ---- prologue: batches 0 .. A in flight; tile 0 landed, converted and published
#pragma unroll
    for (int b = 0; b <= A; ++b) stage(b);
    {
        int younger = nt - 1;
        younger = younger < 0 ? 0 : (younger > A ? A : younger);
        wait_vm_dyn(nb * younger);
        convert(0);
    }
    const uint32_t lane_row = lds0 + Lds::ring + r * kRowBytes + ((h ^ swz(r)) << 4);
    auto frag = [&](uint32_t tb, int ks) {
        return *reinterpret_cast<SVPS_LDS const f16x8*>((uintptr_t)((tb ^ ((ks & 7) << 5)) + 256 * (ks >> 3)));
    };
    constexpr int kOrd[4] = {0, 8, 1, 9};
    int ts = strip0, ty = row0;
    for (int it = 0; it <= nt; ++it) {
        wg_barrier();                                                // B(it): tile it is fp16; the statistics of tile it-1 are in LDS
        combine(it - 1);
        if (it == nt) break;
        const uint32_t tb = lane_row + (uint32_t)(it % NF) * kTileBytes;
        f16x8 kf[2][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) kf[0][u] = frag(tb, kOrd[u]);
        // Cy row and rstd_k of the tile (staged with it)
        f32x16 s;
        {
            const float* cyl = reinterpret_cast<const float*>(smem + Lds::yring + (it % NF) * 1024) + slot0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 cyv = *reinterpret_cast<const f32x4*>(cyl + 8 * g);
#pragma unroll
                for (int j = 0; j < 4; ++j) s[4 * g + j] = cyv[j] * cy_on + cxv[g][j];
            }
        }
        const float rk = *reinterpret_cast<const float*>(smem + Lds::kring + (it % NF) * 256 + r * 4) * kLog2e;
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            if (grp < 3) {
#pragma unroll
                for (int u = 0; u < 4; ++u) kf[(grp + 1) & 1][u] = frag(tb, 2 * (grp + 1) + kOrd[u]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qfh[2 * grp + kOrd[u]], kf[grp & 1][u], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qfl[2 * grp + kOrd[u]], kf[grp & 1][u], s, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        stage(it + A + 1);
        if constexpr (HL) {                                          // fold [Q''.f_hi + Cy + Cx | Q''.f_lo]: both halves then hold the full sum
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(s[i]), __float_as_uint(s[i]), false, false);
                s[i] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            }
        }
        float mloc = kNegBig;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s[4 * g + j] = fmaf(rk, s[4 * g + j], c3v[g][j]);    // padded rows: c3' = -1e30
                mloc = fmaxf(mloc, s[4 * g + j]);
            }
        mloc = ra_half_swap_max(mloc);
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = __builtin_amdgcn_exp2f(s[i] - mloc);
        float sl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) sl[i] = (s[i] + s[4 + i]) + (s[8 + i] + s[12 + i]);
        const float sloc = ra_half_swap_sum((sl[0] + sl[1]) + (sl[2] + sl[3]));
        if (h == 0) stats[((it & 1) * 8 + w) * 32 + r] = make_float2(mloc, sloc);
        // tile it+1: this wave's pieces landed -> fp16. Younger operations than its batch: the batches it+2 .. it+A+1 and the
        // stores of combine() of the iterations in between
        if (it >= A + 1 && it + A + 1 < nt) wait_vm_dyn(A * (nb + 1));
        else wait_vm<0>();
        convert(it + 1);
        ++ty;
        if (ty == H) { ty = 0; ++ts; if (it + 1 < nt) load_cx(ts); }
    }
}

// ============================== more than 128 slots: the two-pass form ===============================================
// (round 3) The softmax runs over ALL slots of a pixel, so a workgroup must see the logits of every slot block before it can
// normalise any of them; Q'' hi / lo for 256 slots plus the accumulators of 256 slots do not fit one CU's registers. The first
// form of this path (kept below: retr_logit_stats_kernel + two EXT launches of retr_attn_kernel) computed every logit TWICE and
// read the map three times. Here the probabilities go through HBM instead:
//   pass 1  retr_probs_kernel   eight waves = eight slot blocks with Q'' hi / lo resident: logits, per-pixel softmax over all 256
//                               rows, P * rstd_v as fp16 -> workspace, one 16-KiB block per tile in EXACTLY the byte layout of
//                               retr_attn_kernel's LDS P tile (slot block sb at sb * 2 KiB, 32 pixel rows of 64 B, chunk swizzle)
//   pass 2  retr_pv_kernel      eight waves = eight slot blocks of accumulators (A[32 sb .., 0:256] + the aux block): feature
//                               tile + P block + aux rows arrive by LDS-DMA, 18 MFMA per wave and tile, no softmax, no Q''
// Matrix work per tile: 256 + 144 MFMAs instead of 256 + 2 x (128 + 72); bytes per pixel: 512 + 512 (P out) in pass 1,
// 512 + 512 + 16 in pass 2. Same operands and roundings as retr_attn_kernel (P * rstd_v is ONE fp16 there as well).
struct ProbsLds {
    static constexpr int kA = 4;                                // tiles ahead
    static constexpr int kNF = kA + 2;
    static constexpr int ring = 0;
    static constexpr int yring = kNF * kTileBytes;              // kNF x 1 KiB Cy rows (256 slots)
    static constexpr int aring = yring + kNF * 1024;            // kNF x 1 KiB aux rows of the tile's pixels
    static constexpr int stats = aring + kNF * kAuxTile;        // [2][8][32] float2
    static constexpr int ptile = stats + 2 * 8 * 32 * 8;        // [8 waves][2 KiB]: this wave's block of the P tile on its way out
    static constexpr int total = ptile + 8 * 2048;
};
static_assert(ProbsLds::total <= 160 * 1024, "LDS layout");
constexpr int kPBlock = 16384;                                  // bytes of P per tile in the workspace: 256 slots x 32 pixels fp16

__global__ __launch_bounds__(512) void retr_probs_kernel(
    const _Float16* __restrict__ qh, const _Float16* __restrict__ ql,  // [T, 256, 256]
    const float* __restrict__ cy, const float* __restrict__ cx,        // [T, H, 256], [T, W, 256]
    const float* __restrict__ c3g,                                     // [T, 256]
    const __bf16* __restrict__ feat, const __bf16* __restrict__ aux,   // aux: the 16-byte rows of retr_stats.hip
    char* __restrict__ pout,                                           // [T, tiles, 16 KiB]
    int HW, int H, int W, int tiles_per_chunk, int map_f16) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    using Lds = ProbsLds;
    constexpr int NF = Lds::kNF, A = Lds::kA, LP = 256;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int t = blockIdx.y, c = blockIdx.x;
    const int tiles = ((W + kTilePx - 1) / kTilePx) * H;
    const int tid0 = c * tiles_per_chunk;
    int nt = tiles - tid0;
    nt = nt < tiles_per_chunk ? nt : tiles_per_chunk;
    const int strip0 = tid0 / H, row0 = tid0 - strip0 * H;
    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>((SVPS_LDS const void*)smem);

    f16x8 qfh[16], qfl[16];
    {
        const size_t row = ((size_t)t * LP + 32 * w + r) * kD + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            qfh[ks] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(qh + row + 16 * ks));
            qfl[ks] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(ql + row + 16 * ks));
        }
    }
    const int slot0 = 32 * w + 4 * h;
    f32x4 c3v[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) c3v[g] = *reinterpret_cast<const f32x4*>(c3g + (size_t)t * LP + slot0 + 8 * g);
    // consume every register loaded above HERE (see retr_logit_stats_kernel)
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) asm volatile("" : "+v"(qfh[ks]), "+v"(qfl[ks]));
#pragma unroll
    for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(c3v[g]));
    wait_vm<0>();

    const u32x4 cys = ra_make_srd(cy + (size_t)t * H * LP, (uint32_t)(H * LP) * 4u);
    const u32x4 cxs = ra_make_srd(cx + (size_t)t * W * LP, (uint32_t)(W * LP) * 4u);
    const u32x4 frs = ra_make_srd(feat + (size_t)t * HW * kD, (uint32_t)HW * kRowBytes);
    const u32x4 ars = ra_make_srd(aux + (size_t)t * HW * 8, (uint32_t)HW * kAuxRow);
    const u32x4 prs = ra_make_srd(pout + ((size_t)t * tiles + tid0) * kPBlock, (uint32_t)nt * (uint32_t)kPBlock);
    auto ld16 = [](u32x4 srd, int off) {                            // asm + its own wait (no compiler-visible load in the loop)
        f32x4 v;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(off), "s"(srd) : "memory");
        return v;
    };
    f32x4 cxv[4];
    auto load_cx = [&](int strip) {
        int xx = kTilePx * strip + r;
        xx = xx < W ? xx : W - 1;
#pragma unroll
        for (int g = 0; g < 4; ++g) cxv[g] = ld16(cxs, (xx * LP + slot0 + 8 * g) * 4);
    };
    load_cx(strip0);

    // ---- staging: rows 4w .. 4w+3 of every tile (two pieces); wave 7: the Cy row; wave 6: the aux rows of the 32 pixels
    int voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 4 * w + 2 * i + h;
        voff[i] = row * kRowBytes + (((lane & 31) ^ swz(row)) * 16);
    }
    const int nb = 2 + (w >= 6 ? 1 : 0);                            // DMA instructions of one batch of this wave
    int ds = strip0, dy = row0;
    auto stage = [&](int tile) {
        if (tile >= nt) return;
        const uint32_t st = __builtin_amdgcn_readfirstlane(lds0 + Lds::ring + (tile % NF) * kTileBytes + w * 2048);
        const int px0 = dy * W + kTilePx * ds;
        const int soff = __builtin_amdgcn_readfirstlane(px0 * kRowBytes);
        if (px0 + kTilePx <= HW) {
            ra_dma16(frs, st, voff[0], soff);
            ra_dma16(frs, st + 1024, voff[1], soff);
        } else {                                                     // last row of a ragged strip: clamp the source rows (their P is 0)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = 4 * w + 2 * i + h;
                const int src = px0 + row < HW ? row : HW - 1 - px0;
                ra_dma16(frs, st + i * 1024, src * kRowBytes + (((lane & 31) ^ swz(row)) * 16), soff);
            }
        }
        if (w == 7) {
            ra_dma16_cached(cys, __builtin_amdgcn_readfirstlane(lds0 + Lds::yring + (tile % NF) * 1024), dy * LP * 4 + lane * 16);
        } else if (w == 6) {                                         // 32 rows of 16 B (lanes >= 32 repeat them); rows past the frame read zeros
            ra_dma16(ars, __builtin_amdgcn_readfirstlane(lds0 + Lds::aring + (tile % NF) * kAuxTile), (px0 + (lane & 31)) * kAuxRow, 0);
        }
        ++dy;
        if (dy == H) { dy = 0; ++ds; }
    };
    auto convert = [&](int tile) {
        if (tile >= nt || map_f16) return;
        const uint32_t st = lds0 + Lds::ring + (tile % NF) * kTileBytes + w * 2048 + lane * 16;
        u32x4 w_[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) w_[i] = *reinterpret_cast<SVPS_LDS const u32x4*>((uintptr_t)(st + i * 1024));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                w_[i][k] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(__uint_as_float(w_[i][k] << 16), __uint_as_float(w_[i][k] & 0xffff0000u)));
            *reinterpret_cast<SVPS_LDS u32x4*>((uintptr_t)(st + i * 1024)) = w_[i];
        }
    };
    float2* stats = reinterpret_cast<float2*>(smem + Lds::stats);
    // P of the previous tile: exponentials relative to this wave's block maximum, the block maximum, rstd_v and "inside the map"
    f32x16 e;
    float mloc_p = 0.f, tau_p = 0.f;
    bool live_p = false;
#pragma unroll
    for (int i = 0; i < 16; ++i) e[i] = 0.f;
    const int key = (r >> 1) & 3;
    char* prow = smem + Lds::ptile + w * 2048 + r * 64 + 8 * h;     // retr_attn_kernel's P tile layout, this wave's block
    const uint32_t pback = lds0 + Lds::ptile + w * 2048 + lane * 16;
    // finish of tile `tile` = it - 1: normalise over the eight blocks, P * rstd_v -> fp16 -> this wave's 2 KiB of the tile's block.
    // The two stores are ALWAYS issued (out-of-range offset when there is nothing to store): the counted vmcnt waits rely on them.
    auto finish = [&](int tile) {
        const bool have = tile >= 0;
        float mall = kNegBig;
        float2 st_w[8];
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) {
            st_w[ww] = stats[((tile & 1) * 8 + ww) * 32 + r];
            mall = fmaxf(mall, st_w[ww].x);
        }
        float den = 0.f;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) den += st_w[ww].y * __builtin_amdgcn_exp2f(st_w[ww].x - mall);
        float fac = __builtin_amdgcn_exp2f(mloc_p - mall) * __builtin_amdgcn_rcpf(den) * tau_p;
        if (!live_p || !have) fac = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f16x4 ph;
#pragma unroll
            for (int j = 0; j < 4; ++j) ph[j] = (_Float16)(e[4 * g + j] * fac);
            *reinterpret_cast<f16x4*>(prow + ((g ^ key) * 16)) = ph;
        }
        // the wave's own LDS operations complete in order: the read-back below sees the writes above
        const u32x4 b0 = *reinterpret_cast<SVPS_LDS const u32x4*>((uintptr_t)pback);
        const u32x4 b1 = *reinterpret_cast<SVPS_LDS const u32x4*>((uintptr_t)(pback + 1024));
        const int so = have ? tile * kPBlock + w * 2048 + lane * 16 : 0x7ffffff0;
        asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen nt" : : "v"(b0), "v"(so), "s"(prs) : "memory");
        asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen offset:1024 nt" : : "v"(b1), "v"(so), "s"(prs) : "memory");
    };

    // ---- prologue: batches 0 .. A in flight; tile 0 landed, converted and published
#pragma unroll
    for (int b = 0; b <= A; ++b) stage(b);
    {
        int younger = nt - 1;
        younger = younger < 0 ? 0 : (younger > A ? A : younger);
        wait_vm_dyn(nb * younger);
        convert(0);
    }
    const uint32_t lane_row = lds0 + Lds::ring + r * kRowBytes + ((h ^ swz(r)) << 4);
    auto frag = [&](uint32_t tb, int ks) {
        return *reinterpret_cast<SVPS_LDS const f16x8*>((uintptr_t)((tb ^ ((ks & 7) << 5)) + 256 * (ks >> 3)));
    };
    constexpr int kOrd[4] = {0, 8, 1, 9};
    int ts = strip0, ty = row0;
    for (int it = 0; it <= nt; ++it) {
        wg_barrier();                                                // B(it): tile it is fp16; the statistics of tile it-1 are in LDS
        finish(it - 1);
        if (it == nt) break;
        const uint32_t tb = lane_row + (uint32_t)(it % NF) * kTileBytes;
        f16x8 kf[2][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) kf[0][u] = frag(tb, kOrd[u]);
        f32x16 s;
        {
            const float* cyl = reinterpret_cast<const float*>(smem + Lds::yring + (it % NF) * 1024) + slot0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 cyv = *reinterpret_cast<const f32x4*>(cyl + 8 * g);
#pragma unroll
                for (int j = 0; j < 4; ++j) s[4 * g + j] = cyv[j] + cxv[g][j];
            }
        }
        // (rstd_k, rstd_v) of this lane's pixel: bytes 8 .. 15 of its aux row
        const f32x2 rt = *reinterpret_cast<SVPS_LDS const f32x2*>((uintptr_t)(lds0 + Lds::aring + (it % NF) * kAuxTile + r * kAuxRow + 8));
        const float rk = rt[0] * kLog2e;
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            if (grp < 3) {
#pragma unroll
                for (int u = 0; u < 4; ++u) kf[(grp + 1) & 1][u] = frag(tb, 2 * (grp + 1) + kOrd[u]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qfh[2 * grp + kOrd[u]], kf[grp & 1][u], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qfl[2 * grp + kOrd[u]], kf[grp & 1][u], s, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        stage(it + A + 1);
        float mloc = kNegBig;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s[4 * g + j] = fmaf(rk, s[4 * g + j], c3v[g][j]);    // padded rows: c3' = -1e30
                mloc = fmaxf(mloc, s[4 * g + j]);
            }
        mloc = ra_half_swap_max(mloc);
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = __builtin_amdgcn_exp2f(s[i] - mloc);
        float sl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) sl[i] = (s[i] + s[4 + i]) + (s[8 + i] + s[12 + i]);
        const float sloc = ra_half_swap_sum((sl[0] + sl[1]) + (sl[2] + sl[3]));
        if (h == 0) stats[((it & 1) * 8 + w) * 32 + r] = make_float2(mloc, sloc);
        e = s;
        mloc_p = mloc;
        tau_p = rt[1] * kPScale;
        live_p = kTilePx * ts + r < W;
        // tile it+1: this wave's pieces landed -> fp16. Younger operations than its batch: the batches it+2 .. it+A+1 and the
        // two stores of finish() of each iteration in between
        if (it >= A + 1 && it + A + 1 < nt) wait_vm_dyn(A * (nb + 2));
        else wait_vm<0>();
        convert(it + 1);
        ++ty;
        if (ty == H) { ty = 0; ++ts; if (it + 1 < nt) load_cx(ts); }
    }
}

struct PvLds {
    static constexpr int kA = 3;                                // batches ahead
    static constexpr int kNF = kA + 1;                          // ring depth: tile it (compute), it+1 (converted), it+2, it+3 in flight
    static constexpr int fring = 0;
    static constexpr int pring = kNF * kTileBytes;
    static constexpr int aring = pring + kNF * kPBlock;
    static constexpr int total = aring + kNF * kAuxTile;
};
static_assert(PvLds::total <= 160 * 1024, "LDS layout");

__global__ __launch_bounds__(512) void retr_pv_kernel(
    const __bf16* __restrict__ feat,    // [T, HW, 256]
    const __bf16* __restrict__ aux,     // [T, HW, 8]
    const char* __restrict__ pin,       // [T, tiles, 16 KiB]  retr_probs_kernel
    float* __restrict__ partial,        // [T, C, L, 260]
    int L, int HW, int H, int W, int tiles_per_chunk, int map_f16) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    using Lds = PvLds;
    constexpr int A = Lds::kA, NF = Lds::kNF;
    const int lane = threadIdx.x & 63;
    const int sb = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // slot block 0 .. 7
    const int r = lane & 31, h = lane >> 5;
    const int C = gridDim.x;
    int t = blockIdx.y, c = blockIdx.x;
    if ((gridDim.y & 7) == 0) {                                  // XCD-aware frame placement (see retr_attn_kernel)
        const int b = blockIdx.y * C + blockIdx.x;
        const int n = b >> 3;
        t = (b & 7) + 8 * (n / C);
        c = n % C;
    }
    const int tiles = ((W + kTilePx - 1) / kTilePx) * H;
    const int tid0 = c * tiles_per_chunk;
    int nt = tiles - tid0;
    nt = nt < tiles_per_chunk ? nt : tiles_per_chunk;
    const int strip0 = tid0 / H, row0 = tid0 - strip0 * H;
    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>((SVPS_LDS const void*)smem);

    const u32x4 frs = ra_make_srd(feat + (size_t)t * HW * kD, (uint32_t)HW * kRowBytes);
    const u32x4 ars = ra_make_srd(aux + (size_t)t * HW * 8, (uint32_t)HW * kAuxRow);
    const u32x4 prs = ra_make_srd(pin + ((size_t)t * tiles + tid0) * kPBlock, (uint32_t)nt * (uint32_t)kPBlock);
    int voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 4 * sb + 2 * i + h;
        voff[i] = row * kRowBytes + (((lane & 31) ^ swz(row)) * 16);
    }
    const int nb = 4 + (sb == 0 ? 1 : 0);                        // DMA instructions of one batch of this wave
    int ds = strip0, dy = row0;
    auto issue_batch = [&](int b) {
        if (b >= nt) return;
        const uint32_t st = __builtin_amdgcn_readfirstlane(lds0 + Lds::fring + (b % NF) * kTileBytes + sb * 2048);
        const int px0 = dy * W + kTilePx * ds;
        const int soff = __builtin_amdgcn_readfirstlane(px0 * kRowBytes);
        if (px0 + kTilePx <= HW) {
            ra_dma16(frs, st, voff[0], soff);
            ra_dma16(frs, st + 1024, voff[1], soff);
        } else {                                                 // last row of a ragged strip: clamp the source rows (their P is 0)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = 4 * sb + 2 * i + h;
                const int src = px0 + row < HW ? row : HW - 1 - px0;
                ra_dma16(frs, st + i * 1024, src * kRowBytes + (((lane & 31) ^ swz(row)) * 16), soff);
            }
        }
        // this wave's 2 KiB of the tile's P block, byte for byte
        const uint32_t sp = __builtin_amdgcn_readfirstlane(lds0 + Lds::pring + (b % NF) * kPBlock + sb * 2048);
        const int poff = __builtin_amdgcn_readfirstlane(b * kPBlock + sb * 2048);
        ra_dma16(prs, sp, lane * 16, poff);
        ra_dma16(prs, sp + 1024, lane * 16 + 1024, poff);
        if (sb == 0) {                                           // aux tile: 32 rows of 16 B (lanes >= 32 repeat them); rows past the frame read zeros
            const uint32_t sa = __builtin_amdgcn_readfirstlane(lds0 + Lds::aring + (b % NF) * kAuxTile);
            ra_dma16(ars, sa, (px0 + (lane & 31)) * kAuxRow, 0);
        }
        ++dy;
        if (dy == H) { dy = 0; ++ds; }
    };
#pragma unroll
    for (int b = 0; b < A; ++b) issue_batch(b);

    f32x16 o[8], oa;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        oa[i] = 0.f;
#pragma unroll
        for (int db = 0; db < 8; ++db) o[db][i] = 0.f;
    }
    // fragment addresses: retr_attn_kernel's consumer, with eight slot blocks in the P tile
    const int g2 = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    const int cl = 2 * (g2 & 1) + (pp >> 1), sub = 8 * (pp & 1), rowl = 8 * (g2 >> 1) + qq;
    const uint32_t lane_v0 = rowl * kRowBytes + (((cl ^ (2 * (g2 >> 1))) + 4 * qq) << 4) + sub;
    const uint32_t lane_v1 = (rowl + 4) * kRowBytes + (((cl ^ (2 * (g2 >> 1) + 1)) + 4 * qq) << 4) + sub;
    const uint32_t lane_p0 = sb * 2048 + sub + rowl * 64 + ((cl ^ (qq >> 1)) << 4);
    const uint32_t lane_p1 = sb * 2048 + sub + (rowl + 4) * 64 + ((cl ^ (qq >> 1) ^ 2) << 4);
    const uint32_t lane_a = rowl * kAuxRow + ((cl == 0 && sub != 0) ? 8 : 0);
    auto tr = [](uint32_t a) {
        return __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16(reinterpret_cast<SVPS_LDS fp16x4_gcc*>((uintptr_t)a)));
    };
    auto cat = [](f16x4 a, f16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); };
    f16x8 ah[2], af[2], vf[2][4];
    uint32_t p0 = 0, p1 = 0, v0 = 0, v1 = 0, aa = 0;
    auto vfrag = [&](int ks, int db) {
        const uint32_t o_ = 8192 * ks + 256 * (db >> 2);
        return cat(tr((v0 ^ ((db & 3) << 6)) + o_), tr((v1 ^ ((db & 3) << 6)) + o_));
    };
    auto pv_begin = [&](int j) {
        const uint32_t pt = lds0 + Lds::pring + (j % NF) * kPBlock, vt = lds0 + Lds::fring + (j % NF) * kTileBytes;
        const uint32_t at = lds0 + Lds::aring + (j % NF) * kAuxTile;
        p0 = pt + lane_p0, p1 = pt + lane_p1, v0 = vt + lane_v0, v1 = vt + lane_v1, aa = at + lane_a;
        ah[0] = cat(tr(p0), tr(p1));
        af[0] = cat(tr(aa), tr(aa + 4 * kAuxRow));
#pragma unroll
        for (int u = 0; u < 4; ++u) vf[0][u] = vfrag(0, u);
    };
    auto pv_rest = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ks = q >> 1, half = q & 1;
            if (q < 3) {
#pragma unroll
                for (int u = 0; u < 4; ++u) vf[(q + 1) & 1][u] = vfrag((q + 1) >> 1, 4 * ((q + 1) & 1) + u);
            }
            if (q == 1) {
                ah[1] = cat(tr(p0 + 1024), tr(p1 + 1024));
                af[1] = cat(tr(aa + 16 * kAuxRow), tr(aa + 20 * kAuxRow));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                o[4 * half + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], vf[q & 1][u], o[4 * half + u], 0, 0, 0);
            if (half == 0) oa = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], af[ks], oa, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto convert_batch = [&](int b) {                            // this wave's two pieces of feature tile b, bf16 -> fp16 in place
        if (b >= nt || map_f16) return;
        const uint32_t st = lds0 + Lds::fring + (b % NF) * kTileBytes + sb * 2048 + lane * 16;
        u32x4 w_[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) w_[i] = *reinterpret_cast<SVPS_LDS const u32x4*>((uintptr_t)(st + i * 1024));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const fp16x2_t pk = __builtin_amdgcn_cvt_pkrtz(__uint_as_float(w_[i][k] << 16), __uint_as_float(w_[i][k] & 0xffff0000u));
                w_[i][k] = __builtin_bit_cast(uint32_t, pk);
            }
            *reinterpret_cast<SVPS_LDS u32x4*>((uintptr_t)(st + i * 1024)) = w_[i];
        }
    };

    for (int it = 0; it < nt; ++it) {
        // batches <= it+1 landed for this wave: all but the A - 2 youngest of the batches 0 .. it+A-1 issued so far
        if (it + A - 1 < nt) wait_vm_dyn(nb * (A - 2));
        else wait_vm<0>();
        if (it == 0) convert_batch(0);
        convert_batch(it + 1);
        wg_barrier();                                            // B(it): tile it is complete (every wave's pieces, fp16) and tile it-1 is free
        pv_begin(it);
        issue_batch(it + A);
        pv_rest();
    }

    float* dst = partial + ((size_t)t * C + c) * L * kPartRow;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int slot = 32 * sb + acc_row(i, h);
        if (slot < L) {
#pragma unroll
            for (int db = 0; db < 8; ++db) dst[(size_t)slot * kPartRow + 32 * db + r] = o[db][i];
            if (r < 4) dst[(size_t)slot * kPartRow + 256 + r] = oa[i];
        }
    }
}

// Sum of the C partials of every (frame, slot) row in chunk order (bitwise reproducible, no float atomics):
// out row (272 floats) = { A[0:256], s1, s0, 0 x 14 }, the operand of the slot-side product with
// [ (gamma_v W~_v)^T ; gamma_v b~_v ; beta_v ; 0 ].
__global__ __launch_bounds__(256) void retr_finish_kernel(const float* __restrict__ partial, float* __restrict__ out, int L, int C) {
    const int l = blockIdx.x, t = blockIdx.y, d = threadIdx.x;
    const size_t cstride = (size_t)L * kPartRow;
    const float* src = partial + ((size_t)t * C * L + l) * kPartRow;
    auto colsum = [&](int col) {
        const float* s = src + col;
        float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int c = 0;
        for (; c + 8 <= C; c += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a8[u] += s[(size_t)(c + u) * cstride];
        }
        for (int u = 0; c < C; ++c, ++u) a8[u] += s[(size_t)c * cstride];
        return ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    };
    float* o = out + ((size_t)t * L + l) * kExtRow;
    o[d] = colsum(d) * kPScaleInv;                           // the probabilities carried 2^7 (common.h)
    if (d < kExtRow - 256) {
        float v = 0.f;
        if (d == 0) v = colsum(256) * kPScaleInv;                         // s1 = sum_p P rstd_v
        else if (d == 1) v = (colsum(257) + colsum(258)) * kPScaleInv;    // s0 = sum_p P  (sigma_v carried as hi + lo)
        o[256 + d] = v;
    }
}

}  // namespace svps

namespace {
struct RetrPlan {
    int chunks, tiles_per_chunk;
};
// retriever: tiles of 32 pixels inside an image row, walked strip by strip (retr_attn_kernel)
RetrPlan plan_retr(int T, int H, int W, int chunks_req, int tpx = svps::kTilePx, int min_tiles = 64) {
    const int tiles = ((W + tpx - 1) / tpx) * H;
    int chunks = chunks_req;
    if (chunks <= 0) chunks = svps_pick_chunks(T, tiles, svps_num_cus(), min_tiles);
    if (chunks > tiles) chunks = tiles;
    const int tpc = (tiles + chunks - 1) / chunks;
    chunks = (tiles + tpc - 1) / tpc;
    return {chunks, tpc};
}
// more than 128 slots: the P blocks of the two-pass form (16 KiB per tile), or the per-pixel statistics of the three-launch form
bool retr_three_launch() {
    static const bool v = getenv("SVPS_RETR_L256_THREE_LAUNCH") != nullptr;      // comparison runs only (tools/kbench_retr.py)
    return v;
}
size_t retr_stats_bytes(int T, int L, int H, int W) {
    if (L <= 128) return 0;
    const size_t tiles = (size_t)((W + svps::kTilePx - 1) / svps::kTilePx) * H;
    return retr_three_launch() ? (size_t)T * H * W * sizeof(float2) : (size_t)T * tiles * svps::kPBlock;
}
}  // namespace

extern "C" size_t svps_retr_attn_workspace_bytes(int T, int L, int H, int W, int chunks) {
    if (T <= 0 || L <= 0 || H <= 0 || W <= 0) return 0;
    const RetrPlan p = plan_retr(T, H, W, chunks);
    return (size_t)T * p.chunks * L * svps::kPartRow * sizeof(float) + retr_stats_bytes(T, L, H, W);
}

extern "C" int svps_retr_attn_fwd(const void* qh, const void* ql, const float* cy, const float* cx, const float* c3,
                                  const void* feat, const void* aux,
                                  void* workspace, size_t workspace_bytes, float* out_ext, int T, int L, int H, int W, int D,
                                  int chunks, int flags, void* stream_) {
    const int mf = (flags & SVPS_FLAG_MAP_F16) ? 1 : 0;              // the map is fp16: the kernels skip their bf16 -> fp16 pass
    if (!qh || !ql || !cy || !cx || !c3 || !feat || !aux || !workspace || !out_ext) return SVPS_ERR_BAD_ARG;
    if (D != svps::kD || T <= 0 || L <= 0 || L > 256 || H <= 0 || W <= 0) return SVPS_ERR_BAD_SHAPE;
    if ((size_t)H * W > svps::kMaxFramePixels) return SVPS_ERR_BAD_SHAPE;
    const int HW = H * W;
    const RetrPlan p = plan_retr(T, H, W, chunks);
    const size_t partial_bytes = (size_t)T * p.chunks * L * svps::kPartRow * sizeof(float);
    if (workspace_bytes < partial_bytes + retr_stats_bytes(T, L, H, W)) return SVPS_ERR_WORKSPACE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    float* partial = static_cast<float*>(workspace);
    const __bf16* qh_ = static_cast<const __bf16*>(qh);
    const __bf16* ql_ = static_cast<const __bf16*>(ql);
    const __bf16* f_ = static_cast<const __bf16*>(feat);
    const __bf16* a_ = static_cast<const __bf16*>(aux);
    hipError_t e;
    svps_prof_mark(SVPS_KERNEL_RETR_ATTN, 0, stream);
    if (L <= 128) {
        auto kern = svps::retr_attn_kernel<0, false>;
        int slot = 0;
#ifdef SVPS_RETR_ABLATE
        // diagnostic build only (tools/ablate.sh builds it as a separate library): timing-only variants that return wrong results
        static const int ablate = [] { const char* a = getenv("SVPS_RETR_ABLATE"); return a ? atoi(a) : 0; }();
        if (ablate == 1) { kern = svps::retr_attn_kernel<1, false>; slot = 1; }
        else if (ablate == 2) { kern = svps::retr_attn_kernel<2, false>; slot = 2; }
        else if (ablate == 4) { kern = svps::retr_attn_kernel<4, false>; slot = 3; }
        else if (ablate == 8) { kern = svps::retr_attn_kernel<8, false>; slot = 4; }
#endif
        static SvpsLdsAttr attr[5];
        if (hipError_t ae = attr[slot].ensure(reinterpret_cast<const void*>(kern), svps::RetrLds::total); ae != hipSuccess) return (int)ae;
        hipLaunchKernelGGL(kern, dim3(p.chunks, T), dim3(512), svps::RetrLds::total, stream, qh_, ql_, cy, cx, c3, f_, a_, partial,
                           L, HW, H, W, p.tiles_per_chunk, 128, L, 0, (const float2*)nullptr, mf, (const __bf16*)nullptr);
        e = hipGetLastError();
    } else if (!retr_three_launch()) {
        // more than 128 slots (padded layouts of 256 rows), two passes: probabilities of all slots -> workspace, then P f
        char* pws = static_cast<char*>(workspace) + partial_bytes;       // partial_bytes is a multiple of 16 (260 floats per row)
        const RetrPlan pl = plan_retr(T, H, W, 0);
        static SvpsLdsAttr attr_p, attr_v;
        if (hipError_t ae = attr_p.ensure(reinterpret_cast<const void*>(svps::retr_probs_kernel), svps::ProbsLds::total); ae != hipSuccess) return (int)ae;
        if (hipError_t ae = attr_v.ensure(reinterpret_cast<const void*>(svps::retr_pv_kernel), svps::PvLds::total); ae != hipSuccess) return (int)ae;
        hipLaunchKernelGGL(svps::retr_probs_kernel, dim3(pl.chunks, T), dim3(512), svps::ProbsLds::total, stream,
                           static_cast<const _Float16*>(qh), static_cast<const _Float16*>(ql), cy, cx, c3, f_, a_, pws, HW, H, W,
                           pl.tiles_per_chunk, mf);
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(svps::retr_pv_kernel, dim3(p.chunks, T), dim3(512), svps::PvLds::total, stream, f_, a_, (const char*)pws,
                           partial, L, HW, H, W, p.tiles_per_chunk, mf);
        e = hipGetLastError();
    } else {
        // the first form of this path (comparison runs): softmax statistics over all slots, then the retriever once per half of
        // the slots with those statistics
        float2* st = reinterpret_cast<float2*>(static_cast<char*>(workspace) + partial_bytes);
        const RetrPlan pl = plan_retr(T, H, W, 0);
        static SvpsLdsAttr attr_s, attr_e;
        if (hipError_t ae = attr_s.ensure(reinterpret_cast<const void*>(svps::retr_logit_stats_kernel<false>), svps::LStatsLds::total); ae != hipSuccess) return (int)ae;
        auto kern = svps::retr_attn_kernel<0, true>;
        if (hipError_t ae = attr_e.ensure(reinterpret_cast<const void*>(kern), svps::RetrLds::total); ae != hipSuccess) return (int)ae;
        hipLaunchKernelGGL(svps::retr_logit_stats_kernel<false>, dim3(pl.chunks, T), dim3(512), svps::LStatsLds::total, stream,
                           static_cast<const _Float16*>(qh), static_cast<const _Float16*>(ql), cy, cx, c3, f_, a_, st, L, HW, H, W,
                           pl.tiles_per_chunk, mf, (const __bf16*)nullptr);
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, dim3(p.chunks, T), dim3(512), svps::RetrLds::total, stream, qh_, ql_, cy, cx, c3, f_, a_, partial,
                           128, HW, H, W, p.tiles_per_chunk, 256, L, 0, (const float2*)st, mf, (const __bf16*)nullptr);
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, dim3(p.chunks, T), dim3(512), svps::RetrLds::total, stream, qh_, ql_, cy, cx, c3, f_, a_, partial,
                           L - 128, HW, H, W, p.tiles_per_chunk, 256, L, 128, (const float2*)st, mf, (const __bf16*)nullptr);
        e = hipGetLastError();
    }
    svps_prof_mark(SVPS_KERNEL_RETR_ATTN, 1, stream);
    if (e != hipSuccess) return (int)e;
    svps_prof_mark(SVPS_KERNEL_RETR_FINISH, 0, stream);
    hipLaunchKernelGGL(svps::retr_finish_kernel, dim3(L, T), dim3(256), 0, stream, partial, out_ext, L, p.chunks);
    svps_prof_mark(SVPS_KERNEL_RETR_FINISH, 1, stream);
    return (int)hipGetLastError();
}

// Reference-precision form of svps_retr_attn_fwd: P * rstd_v as fp16 hi + lo, the map as fp16 hi + lo planes in 16-pixel tiles
// (retr_attn_kernel<0, false, true, true>). More than 128 slots (round 5): the softmax statistics over all 256 slot rows first
// (retr_logit_stats_kernel<true>: 8 B per pixel into the workspace), then the retriever once per half of the slots with those statistics
// (retr_attn_kernel<0, true, true, true>) - every logit is computed twice, nothing of size [L, HW] passes through HBM.
namespace {
// round 6: 32-pixel tiles (retr_attn_hl32.hip) unless SVPS_K1_HL16 asks for the sixteen-pixel form of round 5 (comparison runs)
bool retr_hl16() {
    static const bool v = getenv("SVPS_K1_HL16") != nullptr;
    return v;
}
RetrPlan plan_retr_hl(int T, int H, int W, int chunks) {
    return retr_hl16() ? plan_retr(T, H, W, chunks, 16) : plan_retr(T, H, W, chunks, svps::kRetrHl32TilePx, 32);
}
int launch_retr_hl(const void* qh, const void* ql, const float* cy, const float* cx, const float* c3, const void* feat,
                   const void* feat_lo, const void* aux, void* workspace, size_t workspace_bytes, float* out_ext, int T, int L,
                   int H, int W, int D, int chunks, void* stream_) {
    if (!qh || !ql || !cy || !cx || !c3 || !feat || !feat_lo || !aux || !workspace || !out_ext) return SVPS_ERR_BAD_ARG;
    if (D != svps::kD || T <= 0 || L <= 0 || L > 256 || H <= 0 || W <= 0) return SVPS_ERR_BAD_SHAPE;
    if ((size_t)H * W > svps::kMaxFramePixels) return SVPS_ERR_BAD_SHAPE;
    const int HW = H * W;
    const RetrPlan p = plan_retr_hl(T, H, W, chunks);
    const size_t partial_bytes = (size_t)T * p.chunks * L * svps::kPartRow * sizeof(float);
    const size_t stats_bytes = L > 128 ? (size_t)T * HW * sizeof(float2) : 0;
    if (workspace_bytes < partial_bytes + stats_bytes) return SVPS_ERR_WORKSPACE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    float* partial = static_cast<float*>(workspace);
    using Lds = svps::RetrLdsHL;
    const __bf16* qh_ = static_cast<const __bf16*>(qh);
    const __bf16* ql_ = static_cast<const __bf16*>(ql);
    const __bf16* fh_ = static_cast<const __bf16*>(feat);
    const __bf16* fl_ = static_cast<const __bf16*>(feat_lo);
    const __bf16* a_ = static_cast<const __bf16*>(aux);
    hipError_t e;
    svps_prof_mark(SVPS_KERNEL_RETR_ATTN, 0, stream);
    if (!retr_hl16()) {
        if (L <= 128) {
            e = (hipError_t)svps::retr_attn_hl32_launch(qh, ql, cy, cx, c3, feat, feat_lo, aux, partial, T, L, H, W, p.chunks, p.tiles_per_chunk,
                                                        128, L, 0, nullptr, stream_);
        } else {
            float2* st = reinterpret_cast<float2*>(static_cast<char*>(workspace) + partial_bytes);     // partial_bytes is a multiple of 16
            const RetrPlan pl = plan_retr(T, H, W, 0, svps::kRetrHl32TilePx, 32);
            e = (hipError_t)svps::retr_logit_stats_hl32_launch(qh, ql, cy, cx, c3, feat, feat_lo, aux, st, T, H, W, pl.chunks, pl.tiles_per_chunk,
                                                               stream_);
            if (e != hipSuccess) return (int)e;
            e = (hipError_t)svps::retr_attn_hl32_launch(qh, ql, cy, cx, c3, feat, feat_lo, aux, partial, T, 128, H, W, p.chunks,
                                                        p.tiles_per_chunk, 256, L, 0, st, stream_);
            if (e != hipSuccess) return (int)e;
            e = (hipError_t)svps::retr_attn_hl32_launch(qh, ql, cy, cx, c3, feat, feat_lo, aux, partial, T, L - 128, H, W, p.chunks,
                                                        p.tiles_per_chunk, 256, L, 128, st, stream_);
        }
    } else if (L <= 128) {
        auto kern = svps::retr_attn_kernel<0, false, true, true>;
        int slot = 0;
#ifdef SVPS_RETR_ABLATE
        // diagnostic build only (tools/ablate.sh builds it as a separate library): timing-only variants that return wrong results
        static const int ablate = [] { const char* a = getenv("SVPS_RETR_ABLATE"); return a ? atoi(a) : 0; }();
        if (ablate == 1) { kern = svps::retr_attn_kernel<1, false, true, true>; slot = 1; }
        else if (ablate == 2) { kern = svps::retr_attn_kernel<2, false, true, true>; slot = 2; }
        else if (ablate == 4) { kern = svps::retr_attn_kernel<4, false, true, true>; slot = 3; }
#endif
        static SvpsLdsAttr attrs[4];
        SvpsLdsAttr& attr = attrs[slot];
        if (hipError_t ae = attr.ensure(reinterpret_cast<const void*>(kern), Lds::total); ae != hipSuccess) return (int)ae;
        hipLaunchKernelGGL(kern, dim3(p.chunks, T), dim3(512), Lds::total, stream, qh_, ql_, cy, cx, c3, fh_, a_, partial, L, HW, H, W,
                           p.tiles_per_chunk, 128, L, 0, (const float2*)nullptr, 1, fl_);
        e = hipGetLastError();
    } else {
        float2* st = reinterpret_cast<float2*>(static_cast<char*>(workspace) + partial_bytes);     // partial_bytes is a multiple of 16
        const RetrPlan pl = plan_retr(T, H, W, 0, 16);
        auto kstat = svps::retr_logit_stats_kernel<true>;
        auto kern = svps::retr_attn_kernel<0, true, true, true>;
        static SvpsLdsAttr attr_s, attr_e;
        if (hipError_t ae = attr_s.ensure(reinterpret_cast<const void*>(kstat), svps::LStatsLds::total); ae != hipSuccess) return (int)ae;
        if (hipError_t ae = attr_e.ensure(reinterpret_cast<const void*>(kern), Lds::total); ae != hipSuccess) return (int)ae;
        hipLaunchKernelGGL(kstat, dim3(pl.chunks, T), dim3(512), svps::LStatsLds::total, stream, static_cast<const _Float16*>(qh),
                           static_cast<const _Float16*>(ql), cy, cx, c3, fh_, a_, st, L, HW, H, W, pl.tiles_per_chunk, 1, fl_);
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, dim3(p.chunks, T), dim3(512), Lds::total, stream, qh_, ql_, cy, cx, c3, fh_, a_, partial, 128, HW, H, W,
                           p.tiles_per_chunk, 256, L, 0, (const float2*)st, 1, fl_);
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, dim3(p.chunks, T), dim3(512), Lds::total, stream, qh_, ql_, cy, cx, c3, fh_, a_, partial, L - 128, HW, H, W,
                           p.tiles_per_chunk, 256, L, 128, (const float2*)st, 1, fl_);
        e = hipGetLastError();
    }
    svps_prof_mark(SVPS_KERNEL_RETR_ATTN, 1, stream);
    if (e != hipSuccess) return (int)e;
    svps_prof_mark(SVPS_KERNEL_RETR_FINISH, 0, stream);
    hipLaunchKernelGGL(svps::retr_finish_kernel, dim3(L, T), dim3(256), 0, stream, partial, out_ext, L, p.chunks);
    svps_prof_mark(SVPS_KERNEL_RETR_FINISH, 1, stream);
    return (int)hipGetLastError();
}
}  // namespace

extern "C" size_t svps_retr_attn_hl_workspace_bytes(int T, int L, int H, int W, int chunks) {
    if (T <= 0 || L <= 0 || L > 256 || H <= 0 || W <= 0) return 0;
    const RetrPlan p = plan_retr_hl(T, H, W, chunks);
    return (size_t)T * p.chunks * L * svps::kPartRow * sizeof(float) + (L > 128 ? (size_t)T * H * W * sizeof(float2) : 0);
}

extern "C" int svps_retr_attn_hl_fwd(const void* qh, const void* ql, const float* cy, const float* cx, const float* c3,
                                     const void* feat_hi, const void* feat_lo, const void* aux, void* workspace, size_t workspace_bytes,
                                     float* out_ext, int T, int L, int H, int W, int D, int chunks, void* stream_) {
    return launch_retr_hl(qh, ql, cy, cx, c3, feat_hi, feat_lo, aux, workspace, workspace_bytes, out_ext, T, L, H, W, D, chunks, stream_);
}

#ifdef SVPS_RETR_STAMP
extern "C" int svps_retr_debug_read(unsigned long long* stamps, unsigned long long* clock) {
    hipError_t e = hipMemcpyFromSymbol(stamps, HIP_SYMBOL(svps::retr_stamps), sizeof(unsigned long long) * 2 * 8 * 8);
    if (e != hipSuccess) return (int)e;
    return (int)hipMemcpyFromSymbol(clock, HIP_SYMBOL(svps::retr_clock), sizeof(unsigned long long) * 4096 * 4);
}
#endif
