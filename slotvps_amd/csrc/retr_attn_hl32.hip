// K1'-HL32 - the statistics-fused retriever (retr_attn.hip; MaskDynamicConv.forward, mmdet/models/detectors/dynamic_mask_head.py:423-461)
// in the reference's precision on THIRTY-TWO-pixel tiles (round 6).
//
// The level map is two fp16 planes (f = hi + lo to 22 bits). The first hi / lo form (retr_attn_kernel<.., HL = true>) kept the 16-KiB
// ring stage of the 16-bit kernels and put sixteen pixels into it - hi rows 0 .. 15, lo rows 16 .. 31 - so that one B fragment carries
// [f_hi | f_lo] of the same pixels. That costs the producers a quarter of their matrix work (Q''_lo . f_lo comes out with Q''_lo . f_hi
// and is worth nothing), a register fold behind every chain, and every per-tile cost (barrier, landing wait, softmax exchange, position
// rows) once per sixteen pixels. Here a ring stage is a hi tile AND a lo tile of the same 32 pixels (32 KiB), as mask_decode_hl32_kernel
// does it:
//   producers   s += Q''_lo f_hi + Q''_hi f_lo + Q''_hi f_hi : 48 MFMAs per 32 pixels (64 before), accumulator columns = pixels, no fold;
//               the softmax head works on 16 registers per lane as in the 16-bit form
//   consumers   A += P_lo f_hi + P_hi f_lo + P_hi f_hi (+ the aux block): 52 MFMAs per 32 pixels (as before), P tiles of 32 pixel rows
// Four stages of 32 KiB leave no room for a second P buffer and a third live stage, so the schedule changes with the tile: the consumers
// run ONE tile behind the producers (not two) and a tile has TWO workgroup barriers (the same number per pixel as before):
//
//   producer, iteration it:   chain(it) || d = rk (Cx + Cy) + c3' | softmax head(it) -> stats | B1(it) | first fragments of it+1; factor,
//                             P(it) -> LDS | B2(it)
//   consumer, iteration it:   P(it-1) fragments; LDS-DMA of batch it+2; A += P(it-1) f(it-1), first part | landing wait of batch it+1 | B1(it)
//                             | rest of A += P(it-1) f(it-1) | B2(it)
//
// P is single-buffered: P(it) is written between B1(it) and B2(it) and read (into registers, at once) after B2(it); the statistics are
// double-buffered; stage (it+2) % 4 was last read by the consumers before B2(it-1). Everything else - tile walk down 32-pixel column
// strips, Cx in registers per strip, Cy rows / aux rows / feature tiles by LDS-DMA, swizzled 512-B pixel rows, transposed value fragments,
// the aux block, partials summed in fixed order by retr_finish_kernel - is the scheme of retr_attn.hip, whose comments carry the derivation.
//
// Measured (finest level 256 x 512, T = 40, 100 slots; profiles/r06/README.md): 1 780 - 1 800 us against 2 090 - 2 110 of the sixteen-pixel
// form on the same box. Timing-only ablations: DMA + barriers 859, producers only 1 370, consumers only 1 170, no LDS-DMA behind the first
// ring fill 1 620, no landing wait 1 780; without the producers' Q''_lo MFMAs (16 of 48) - 170 us, without the consumers' P_lo MFMAs (16 of
// 52) - 195 us, without both - 375: every MFMA costs its 32 cycles of the SIMD's matrix pipe, which the producer and the consumer wave of a
// SIMD share (100 MFMAs = 3 200 of ~4 500 cycles per tile); the rest is the part of the producers' vector work (head + finish) the
// consumers' MFMAs do not cover, and two barrier latencies. Built and measured equal or slower on the same box, not kept
// (profiles/r06/k1hl32_experiment_options.patch): the hand-over of P(it) through one LDS flag per slot block instead of B2 (+ 7 %: the
// consumer then starts its tile late and becomes the critical wave), two alternating accumulators in the chain and / or s_setprio around
// it, the next group's reads forced in front of a group's MFMAs (sched_group_barrier), halving the finish's vector instructions
// (v_cvt_pk_f16_f32 + v_fma_mixlo / mixhi_f16: kept, not faster), the consumers' split point 3 ... 7 (4).
// What DID pay after that (-5 %, 1 855 -> 1 765 us on one box, 1 780 -> 1 675 on another): the consumers' value fragments three sub-steps
// ahead instead of one - sub-steps of ONE channel block (3 MFMAs) with four fragment buffers in the registers of two buffers of two blocks:
// a consumer MFMA cost ~41 cycles, not 32, because a fragment requested 192 cycles ahead is not there under load. Tried after that and
// not kept (profiles/r06/k1hl32_late_experiment_options.patch): the same for the producers (one k-step per buffer, three ahead) + 1.8 %; a
// third barrier behind the chain (strict ping-pong) equal; the nine LDS-DMA pieces of a batch between the consumers' sub-steps + 9.5 %, the
// batch in front of B2 instead of behind it + 10 %: an LDS-DMA instruction next to in-flight MFMAs of its wave stalls them.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "retr_common.h"
#include "../../include/slotvps_hip.h"

namespace svps {

typedef __attribute__((ext_vector_type(2))) __fp16 fp16x2_t;
typedef __fp16 fp16x4_gcc __attribute__((__vector_size__(4 * sizeof(__fp16))));

#ifndef SVPS_RETR_HL32_PREFETCH
#define SVPS_RETR_HL32_PREFETCH 2
#endif
#ifndef SVPS_RETR_HL32_SUB
#define SVPS_RETR_HL32_SUB 1        // consumers: channel blocks per sub-step (1: four fragment buffers, three sub-steps ahead; 2: two buffers, one ahead)
#endif
#ifndef SVPS_RETR_HL32_SPLIT
#define SVPS_RETR_HL32_SPLIT 4      // consumer steps (of 8) in front of B1
#endif

struct RetrLdsHL32 {
    static constexpr int kA = SVPS_RETR_HL32_PREFETCH;         // batches requested ahead
    static constexpr int kNF = kA + 2;                          // tiles it-1 .. it+A are live in iteration it
    static constexpr int kStage = 2 * kTileBytes;               // hi tile, lo tile
    static constexpr int fring = 0;
    static constexpr int aring = kNF * kStage;
    static constexpr int yring = aring + kNF * kAuxTile;
    static constexpr int pring = yring + kNF * kCyTile;         // P hi [8 KiB], P lo [8 KiB]: slot block sb at sb * 2 KiB, 32 pixel rows of 64 B
    static constexpr int stats = pring + 2 * kPTile;            // [2][4][32] float2
    static constexpr int c3 = stats + 2 * 4 * 32 * 8;           // [128] float
    static constexpr int total = c3 + 128 * 4;
};
static_assert(RetrLdsHL32::pring % 512 == 0 && RetrLdsHL32::total <= 160 * 1024, "LDS layout");

#ifdef SVPS_RETR_STAMP
// diagnostic build only (tools/retr32_stamps.py): s_memtime stamps of one workgroup's producer 0 and consumer 0, iterations 8 .. 15
__device__ unsigned long long retr32_stamps[2][8][8];      // [producer / consumer][iteration - 8][point]
#define R32_STAMP(role, pt)                                                                           \
    do {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        if (blockIdx.x == 3 && blockIdx.y == 2 && sb == 0 && it >= 8 && it < 16 && lane == 0)         \
            retr32_stamps[role][it - 8][pt] = __builtin_amdgcn_s_memtime();                           \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    } while (0)
#else
#define R32_STAMP(role, pt) do {} while (0)
#endif

// ABL: timing-only ablations (diagnostic builds), outputs wrong. 1: DMA + barriers only  2: producers only  4: consumers only
//      8: no LDS-DMA behind the first ring fill (everything else runs on stale tiles)  16: no landing wait
template <int ABL, bool EXT>
__global__ __launch_bounds__(512) void retr_attn_hl32_kernel(
    const _Float16* __restrict__ qh,    // [T, LP, 256]  hi(Q''), rows >= the real slot count zero
    const _Float16* __restrict__ ql,    // [T, LP, 256]  lo(Q'')
    const float* __restrict__ cy,       // [T, H, LP]
    const float* __restrict__ cx,       // [T, W, LP]
    const float* __restrict__ c3g,      // [T, LP]
    const _Float16* __restrict__ feat,  // [T, HW, 256] hi plane
    const _Float16* __restrict__ feat_lo,
    const _Float16* __restrict__ aux,   // [T, HW, 8]
    float* __restrict__ partial,        // [T, C, Lrow, 260]
    int L, int HW, int H, int W, int tiles_per_chunk, int LP, int Lrow, int slot_off, const float2* __restrict__ ext_stats) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    using Lds = RetrLdsHL32;
    constexpr int A = Lds::kA;
    constexpr int NF = Lds::kNF;
    constexpr int TPX = 32;
    constexpr int kPLo = kPTile;

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sb = w & 3;
    const bool consumer = w >= 4;
    const int r = lane & 31, h = lane >> 5;
    const int C = gridDim.x;
    int t = blockIdx.y, c = blockIdx.x;
    if ((gridDim.y & 7) == 0) {                                 // XCD-aware frame placement (retr_attn.hip)
        const int b = blockIdx.y * C + blockIdx.x;
        const int n = b >> 3;
        t = (b & 7) + 8 * (n / C);
        c = n % C;
    }
    const int tiles = ((W + TPX - 1) / TPX) * H;
    const int tid0 = c * tiles_per_chunk;
    int nt = tiles - tid0;
    nt = nt < tiles_per_chunk ? nt : tiles_per_chunk;           // >= 1 by construction of the grid
    const int strip0 = tid0 / H, row0 = tid0 - strip0 * H;
    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>((SVPS_LDS const void*)smem);

    float* c3l = reinterpret_cast<float*>(smem + Lds::c3);
    if (threadIdx.x < 128) c3l[threadIdx.x] = c3g[(size_t)t * LP + slot_off + threadIdx.x];

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
        const uint32_t lane_row = lds0 + Lds::fring + r * kRowBytes + ((h ^ swz(r)) << 4);
        auto uniform_rsrc = [](const void* p, int bytes) {
            const uint64_t a = reinterpret_cast<uint64_t>(p);
            const uint64_t u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32)) << 32) |
                               (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a);
            return __builtin_amdgcn_make_buffer_rsrc((void*)u, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
        };
        const __amdgpu_buffer_rsrc_t cxr = uniform_rsrc(cx + (size_t)t * W * LP + slot_off, W * LP * 4 - slot_off * 4);
        const __amdgpu_buffer_rsrc_t exr = uniform_rsrc(EXT ? (const void*)(ext_stats + (size_t)t * HW) : (const void*)cy, EXT ? HW * 8 : 0);

        // Cx of this lane's pixel column: constant down a strip (clamped past the right edge: those pixels are masked); added BEHIND the
        // chain - the accumulator starts from the Cy row alone, read from LDS when the chain starts (no register copy of Cy + Cx)
        f32x4 cxv[4];
        auto load_cx = [&](int strip) {
            int xx = TPX * strip + r;
            xx = xx < W ? xx : W - 1;
            const int xo = (xx * LP + slot0) * 4;
#pragma unroll
            for (int g = 0; g < 4; ++g) cxv[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(cxr, xo + 32 * g, 0, 0));
        };
        int ts = strip0, ty = row0;                                 // strip / image row of tile `it`
        f32x2 ext_n = {0.f, 0.f};                                   // EXT: statistics of this lane's pixel, requested one tile ahead
        auto request_ext = [&](int strip, int row) {
            if constexpr (EXT) {
                int px = row * W + TPX * strip + r;
                px = px < HW ? px : HW - 1;
                ext_n = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(exr, px * 8, 0, 0));
            }
        };
        load_cx(ts);
        request_ext(ts, ty);
        float2* stats = reinterpret_cast<float2*>(smem + Lds::stats);

        // LDS byte address of this lane's 16-B chunk of k-step ks = 8 a + b in a tile: (tile + lane_row) ^ (b << 5), + 256 a; lo tile + 16 KiB
        auto frag = [&](uint32_t tb, int ks) {
            return *reinterpret_cast<SVPS_LDS const f16x8*>((uintptr_t)((tb ^ ((ks & 7) << 5)) + 256 * (ks >> 3)));
        };
        // fragment group g: k-steps g and g + 8 of the hi tile, then of the lo tile (one address, four immediate offsets)
        f16x8 kf[2][4];
        auto load_grp = [&](int buf, uint32_t tb, int g) {
            kf[buf][0] = frag(tb, g);
            kf[buf][1] = frag(tb, g + 8);
            kf[buf][2] = frag(tb + kTileBytes, g);
            kf[buf][3] = frag(tb + kTileBytes, g + 8);
        };
        f32x2 rt = {0.f, 0.f};
        auto prefetch = [&](int tile) {                             // first fragment group and (rstd_k, rstd_v) of a tile
            const uint32_t slot = (uint32_t)(tile % NF);
            load_grp(0, lane_row + slot * Lds::kStage, 0);
            rt = *reinterpret_cast<SVPS_LDS const f32x2*>((uintptr_t)(lds0 + Lds::aring + slot * kAuxTile + r * kAuxRow + 8));
        };

        constexpr bool kRun = ABL != 1 && ABL != 4;
        wg_barrier();                                               // B(start): batch 0 and the Cy row of tile 0 landed
        if constexpr (kRun) prefetch(0);
        for (int it = 0; it < nt; ++it) {
            float fac = 0.f;
            const bool more = it + 1 < nt;
            R32_STAMP(0, 0);
            if constexpr (kRun) {
                const uint32_t tb = lane_row + (uint32_t)(it % NF) * Lds::kStage;
                const float rk_c = rt[0] * kLog2e, tau_c = rt[1] * kPScale;      // common.h: the probabilities carry 2^7
                // d = (log2(e) rstd_k) (Cx + Cy) + c3', the affine part of the logits: from the tile's Cy row (staged one batch early) and the c3
                // terms in the shadow of the chain's first groups - the head is then ONE fma per logit, and the accumulator starts from zero
                const float* cyl = reinterpret_cast<const float*>(smem + Lds::yring + (it % NF) * kCyTile) + slot_off + slot0;
                f32x4 d[4], cyq, c3q;
                f32x16 s;
#pragma unroll
                for (int i = 0; i < 16; ++i) s[i] = 0.f;
                // ---- chain: eight groups of two k-steps, three MFMAs per k-step; the reads of group g + 1 in the shadow of group g
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    if (g < 7) load_grp((g + 1) & 1, tb, g + 1);
                    if (g >= 1 && g < 5) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            d[g - 1][j] = fmaf(rk_c, cxv[g - 1][j] + cyq[j], c3q[j]);
                            asm volatile("" : "+v"(d[g - 1][j]));       // HERE, under this group's MFMAs (hipcc otherwise sinks it into the head)
                        }
                    }
                    if (g < 4) {
                        cyq = *reinterpret_cast<const f32x4*>(cyl + 8 * g);
                        c3q = *reinterpret_cast<const f32x4*>(c3l + slot0 + 8 * g);
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int ks = g + 8 * u;
#ifndef SVPS_R32_SKIP_QLO                                              // (timing experiments only)
                        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qfl[ks], kf[g & 1][u], s, 0, 0, 0);
#endif
                        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qfh[ks], kf[g & 1][2 + u], s, 0, 0, 0);
                        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qfh[ks], kf[g & 1][u], s, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                R32_STAMP(0, 1);
                // ---- softmax head: log2(e) * S = (log2(e) rstd_k) * (Q''.f + Cy + Cx) + c3'. Rows past the real slot count need no
                // masking: their Q'', Cy, Cx are zero and their c3' is -1e30 (retr_query_prep): they exp2 to exactly 0
                const bool live = TPX * ts + r < W;
                ++ty;
                float mloc = kNegBig;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        s[4 * g + j] = fmaf(rk_c, s[4 * g + j], d[g][j]);
                        if constexpr (!EXT) mloc = fmaxf(mloc, s[4 * g + j]);
                    }
                }
                if (ty == H) { ty = 0; ++ts; if (more) load_cx(ts); }
                const f32x2 ext_c = ext_n;
                if constexpr (EXT) mloc = ext_c[0];                 // statistics over all slots are known (log2 domain as well)
                else mloc = ra_half_swap_max(mloc);
                if (more) request_ext(ts, ty);
#pragma unroll
                for (int i = 0; i < 16; ++i) s[i] -= mloc;
#pragma unroll
                for (int i = 0; i < 16; ++i) s[i] = __builtin_amdgcn_exp2f(s[i]);
                if constexpr (!EXT) {
                    float sl[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) sl[i] = (s[i] + s[4 + i]) + (s[8 + i] + s[12 + i]);
                    float sloc = (sl[0] + sl[1]) + (sl[2] + sl[3]);
                    sloc = ra_half_swap_sum(sloc);
                    float2* st = stats + (it & 1) * 128;            // double-buffered: nothing orders the producers between two B1
                    if (h == 0) st[sb * 32 + r] = make_float2(mloc, sloc);
                    R32_STAMP(0, 2);
                    wg_barrier();                                   // B1(it)
                    R32_STAMP(0, 3);
                    float2 st_w[4];
#pragma unroll
                    for (int ww = 0; ww < 4; ++ww) st_w[ww] = st[ww * 32 + r];
                    if (more) prefetch(it + 1);                     // batch it+1 landed before B1: its first fragments under the finish
                    float mall = kNegBig;
#pragma unroll
                    for (int ww = 0; ww < 4; ++ww) mall = fmaxf(mall, st_w[ww].x);
                    float den = 0.f;
#pragma unroll
                    for (int ww = 0; ww < 4; ++ww) den += st_w[ww].y * __builtin_amdgcn_exp2f(st_w[ww].x - mall);
                    fac = __builtin_amdgcn_exp2f(mloc - mall) * __builtin_amdgcn_rcpf(den) * tau_c;
                } else {
                    wg_barrier();                                   // B1(it)
                    if (more) prefetch(it + 1);
                    fac = ext_c[1] * tau_c;
                }
                if (!live) fac = 0.f;                               // pixels past the right edge of the map
                // ---- finish: P(it) = e * fac as fp16 hi + lo into the P tiles (row = pixel r, 64 B = 32 slots of this block); the consumers'
                // reads of P(it-1) completed before they entered B1(it)
                char* prow = smem + Lds::pring + sb * 2048 + r * 64 + 8 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f16x4 ph, pl;
                    // hi = rne(e fac) two at a time (v_cvt_pk_f16_f32), lo = rne(e fac - hi) straight from the fused product
                    // (v_fma_mixlo / mixhi_f16: the fp32 fma, the conversion and the half-register write in one instruction)
#pragma unroll
                    for (int j = 0; j < 4; j += 2) {
                        typedef __attribute__((ext_vector_type(2))) float f32x2_;
                        typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_;
                        const f32x2_ x = {s[4 * g + j] * fac, s[4 * g + j + 1] * fac};
                        const f16x2_ hp = __builtin_convertvector(x, f16x2_);
                        ph[j] = hp[0];
                        ph[j + 1] = hp[1];
                        pl[j] = (_Float16)fmaf(s[4 * g + j], fac, -(float)hp[0]);
                        pl[j + 1] = (_Float16)fmaf(s[4 * g + j + 1], fac, -(float)hp[1]);
                    }
                    *reinterpret_cast<f16x4*>(prow + ((g ^ key) * 16)) = ph;
                    *reinterpret_cast<f16x4*>(prow + kPLo + ((g ^ key) * 16)) = pl;
                }
                R32_STAMP(0, 4);
            } else {
                wg_barrier();                                       // B1(it)
            }
            wg_barrier();                                           // B2(it)
        }
        return;
    }

    // =================================== consumer ===============================================
    // waves 0, 1 stage the hi tile (sixteen rows each), waves 2, 3 the lo tile
    const int pl_ = sb >> 1, rb = 16 * (sb & 1);
    const u32x4 frs = ra_make_srd((pl_ ? feat_lo : feat) + (size_t)t * HW * kD, (uint32_t)HW * kRowBytes);
    const u32x4 ars = ra_make_srd(aux + (size_t)t * HW * 8, (uint32_t)HW * kAuxRow);
    const u32x4 yrs = ra_make_srd(cy + (size_t)t * H * LP, (uint32_t)(H * LP) * 4u);
    int voff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = rb + 2 * i + h;                          // row of the LDS tile
        voff[i] = row * kRowBytes + (((lane & 31) ^ swz(row)) * 16);
    }
    const int nb = 8 + ((sb == 0 || sb == 2) ? 1 : 0);           // DMA instructions of one batch of this wave
    int ds = strip0, dy = row0;                                  // strip / image row of the next batch
    // a batch = nine 1-KiB pieces of this wave (eight feature pieces, the aux tile or the Cy row), issued in a row in front of the tile's MFMAs
    // (spread between the consumers' sub-steps, or issued in front of B2, they measured 10 % slower: profiles/r06/README.md)
    auto issue_batch = [&](int b) {
        if (b >= nt) return;
        if (ABL == 8 && b >= NF) return;                         // timing only: no memory traffic behind the first ring fill
        const uint32_t st = __builtin_amdgcn_readfirstlane(lds0 + Lds::fring + (b % NF) * Lds::kStage + pl_ * kTileBytes + rb * kRowBytes);
        const int px0 = dy * W + TPX * ds;
        const int soff = __builtin_amdgcn_readfirstlane(px0 * kRowBytes);
        if (px0 + TPX <= HW) {
#pragma unroll
            for (int i = 0; i < 8; ++i) ra_dma16(frs, st + i * 1024, voff[i], soff);
        } else {                                                 // last row of a ragged strip: clamp the source rows (their P is 0)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = rb + 2 * i + h;
                const int src = px0 + row < HW ? row : HW - 1 - px0;
                ra_dma16(frs, st + i * 1024, src * kRowBytes + (((lane & 31) ^ swz(row)) * 16), soff);
            }
        }
        ++dy;
        if (dy == H) { dy = 0; ++ds; }
        if (sb == 0) {                                           // aux tile: 32 rows of 16 B (lanes >= 32 repeat them); rows past the frame read zeros
            const uint32_t sa = __builtin_amdgcn_readfirstlane(lds0 + Lds::aring + (b % NF) * kAuxTile);
            ra_dma16(ars, sa, (px0 + (lane & 31)) * kAuxRow, 0);
        } else if (sb == 2) {                                    // Cy row of tile b + 1 (1 KiB from the start of its image row)
            const uint32_t sy = __builtin_amdgcn_readfirstlane(lds0 + Lds::yring + ((b + 1) % NF) * kCyTile);
            ra_dma16_cached(yrs, sy, dy * LP * 4 + lane * 16);
        }
    };
    if (sb == 2) ra_dma16_cached(yrs, __builtin_amdgcn_readfirstlane(lds0 + Lds::yring), row0 * LP * 4 + lane * 16);   // Cy row of tile 0
#pragma unroll
    for (int b = 0; b < A; ++b) issue_batch(b);

    f32x16 o[8], oa;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        oa[i] = 0.f;
#pragma unroll
        for (int db = 0; db < 8; ++db) o[db][i] = 0.f;
    }
    // Fragment addresses as in retr_attn.hip:
    //   value  V[pixels 16 ks + 8 h' .. + 8][channel 32 db + n]: vt + 8192 ks + 256 (db >> 2) + (lane_v{0,1} ^ ((db & 3) << 6)); lo tile + 16 KiB
    //   P      rows rowl (+4) of 64 B, chunk cl ^ ((row >> 1) & 3): pt + 1024 ks + lane_p{0,1}; lo + 8 KiB
    //   aux    rows rowl (+4) of 16 B, linear: at + 256 ks + lane_a
    const int g2 = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    const int cl = 2 * (g2 & 1) + (pp >> 1), sub = 8 * (pp & 1), rowl = 8 * (g2 >> 1) + qq;
    const uint32_t lane_v0 = rowl * kRowBytes + (((cl ^ (2 * (g2 >> 1))) + 4 * qq) << 4) + sub;
    const uint32_t lane_v1 = (rowl + 4) * kRowBytes + (((cl ^ (2 * (g2 >> 1) + 1)) + 4 * qq) << 4) + sub;
    const uint32_t p0 = lds0 + Lds::pring + sb * 2048 + sub + rowl * 64 + ((cl ^ (qq >> 1)) << 4);
    const uint32_t p1 = lds0 + Lds::pring + sb * 2048 + sub + (rowl + 4) * 64 + ((cl ^ (qq >> 1) ^ 2) << 4);
    const uint32_t lane_a = rowl * kAuxRow + ((cl == 0 && sub != 0) ? 8 : 0);
    auto tr = [](uint32_t a) {
        return __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16(reinterpret_cast<SVPS_LDS fp16x4_gcc*>((uintptr_t)a)));
    };
    auto cat = [](f16x4 a, f16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); };
    // A += P f in sub-steps of kSB channel blocks (k-step = pixels 0 .. 15 / 16 .. 31 first, then the blocks); NVB value-fragment buffers: the
    // fragments of sub-steps q + 1 .. q + NVB - 1 are in flight beside sub-step q. A "step" of the split points = two blocks (six MFMAs).
    constexpr int kSB = SVPS_RETR_HL32_SUB;                     // channel blocks per sub-step: 2 (two buffers) or 1 (four buffers, the same registers)
    constexpr int NVB = 4 / kSB;
    constexpr int kNQ = 16 / kSB;                               // sub-steps per tile
    f16x8 ah[2], al[2], af[2], vh[NVB][kSB], vl[NVB][kSB];
    uint32_t v0 = 0, v1 = 0;
    auto vfrag = [&](int lo, int ks, int db) {
        const uint32_t o_ = kTileBytes * lo + 8192 * ks + 256 * (db >> 2);
        return cat(tr((v0 ^ ((db & 3) << 6)) + o_), tr((v1 ^ ((db & 3) << 6)) + o_));
    };
    auto load_step = [&](int buf, int q) {
        const int ks = q / (8 / kSB), db = kSB * (q % (8 / kSB));
#pragma unroll
        for (int u = 0; u < kSB; ++u) {
            vh[buf][u] = vfrag(0, ks, db + u);
            vl[buf][u] = vfrag(1, ks, db + u);
        }
    };
    auto pv_begin = [&](int j) {                                 // every fragment of P(j) (the P tiles are free again at B1), the aux rows, the first sub-steps
        const uint32_t vt = lds0 + Lds::fring + (j % NF) * Lds::kStage;
        const uint32_t aa = lds0 + Lds::aring + (j % NF) * kAuxTile + lane_a;
        v0 = vt + lane_v0, v1 = vt + lane_v1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            ah[ks] = cat(tr(p0 + 1024 * ks), tr(p1 + 1024 * ks));
            al[ks] = cat(tr(p0 + kPLo + 1024 * ks), tr(p1 + kPLo + 1024 * ks));
            af[ks] = cat(tr(aa + 16 * ks * kAuxRow), tr(aa + (16 * ks + 4) * kAuxRow));
        }
#pragma unroll
        for (int q = 0; q < NVB - 1; ++q) load_step(q, q);
    };
    auto pv_steps = [&](auto lo_tag, auto hi_tag) {              // steps [Q0, Q1) of A += P f: six MFMAs each (+ two of the aux block per k-step)
        constexpr int Q0 = decltype(lo_tag)::value * (2 / kSB), Q1 = decltype(hi_tag)::value * (2 / kSB);
#pragma unroll
        for (int q = Q0; q < Q1; ++q) {
            const int ks = q / (8 / kSB), db = kSB * (q % (8 / kSB));
            if (q + NVB - 1 < kNQ) load_step((q + NVB - 1) % NVB, q + NVB - 1);
            __builtin_amdgcn_sched_barrier(0);
#ifndef SVPS_R32_SKIP_PLO                                              // (timing experiments only)
#pragma unroll
            for (int u = 0; u < kSB; ++u) o[db + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], vh[q % NVB][u], o[db + u], 0, 0, 0);
#endif
#pragma unroll
            for (int u = 0; u < kSB; ++u) o[db + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], vl[q % NVB][u], o[db + u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < kSB; ++u) o[db + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], vh[q % NVB][u], o[db + u], 0, 0, 0);
            if (db == 0) {
                oa = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], af[ks], oa, 0, 0, 0);
                oa = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], af[ks], oa, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    constexpr int kSplit = SVPS_RETR_HL32_SPLIT;
    using I0 = std::integral_constant<int, 0>;
    using IS = std::integral_constant<int, kSplit>;
    using I8 = std::integral_constant<int, 8>;
    constexpr bool kWork = ABL != 1 && ABL != 2;

    if (A - 1 < nt) wait_vm_dyn(nb * (A - 1));                   // batch 0 (and the Cy row of tile 0) landed
    else wait_vm<0>();
    wg_barrier();                                                // B(start)
    for (int it = 0; it < nt; ++it) {
        const bool work = kWork && it >= 1;
        R32_STAMP(1, 0);
        if (work) pv_begin(it - 1);
        issue_batch(it + A);
        if (work) pv_steps(I0{}, IS{});
        R32_STAMP(1, 1);
        // batch it+1 landed for this wave (the producers read its first fragments behind B1): all but the A - 1 youngest batches
        if (ABL != 16) {
            if (it + A < nt) wait_vm_dyn(nb * (A - 1));
            else wait_vm<0>();
        } else if (it + 4 * A < nt) wait_vm_dyn(nb * (4 * A - 1));  // (timing only: the landing wait never binds)
        R32_STAMP(1, 2);
        wg_barrier();                                            // B1(it)
        R32_STAMP(1, 3);
        if (work) pv_steps(IS{}, I8{});
        R32_STAMP(1, 4);
        wg_barrier();                                            // B2(it)
    }
    if (kWork) {
        pv_begin(nt - 1);
        pv_steps(I0{}, I8{});
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

// Per-pixel softmax statistics over up to 256 slots in the reference's precision, on the 32-pixel hi / lo tiles of the retriever above
// (first kernel of the path for more than 128 slots; retr_logit_stats_kernel<true> in retr_attn.hip is the sixteen-pixel form):
// stats[t, p] = (max_l S[l, p], 1 / sum_l exp2(S[l, p] - max)) with S exactly as retr_attn_hl32_kernel computes it (same fp16 operands,
// same strip tile order, same position-term handling). Eight waves = eight slot blocks of 32 rows, Q'' hi / lo resident (128 registers),
// 48 MFMAs per wave and tile (64 per 32 pixels in the sixteen-pixel form). Every wave stages four 1-KiB pieces of every stage (waves 0 .. 3
// the hi tile, 4 .. 7 the lo tile; LDS-DMA, two stages ahead in a ring of four); wave 7 also stages the Cy row of the tile (1 KiB = 256
// slots), wave 6 the 32 rstd_k values. ONE barrier per tile: the statistics of the eight blocks are double-buffered and combined at the
// START of the next iteration (wave w: pixels 4 w .. 4 w + 3). Reads the planes once, writes 8 B per pixel.
struct LStatsLdsHL32 {
    static constexpr int kA = 2;                                // stages ahead
    static constexpr int kNF = kA + 2;                          // tile it (compute), it+1 (landed), it+2 .. it+A+1 in flight
    static constexpr int kStage = 2 * kTileBytes;
    static constexpr int ring = 0;
    static constexpr int yring = kNF * kStage;                  // kNF x 1 KiB Cy rows
    static constexpr int kring = yring + kNF * 1024;            // kNF x 256 B rstd_k of the tile's pixels (64 lanes x 4 B, lanes >= 32 repeat)
    static constexpr int stats = kring + kNF * 256;             // [2][8][32] float2
    static constexpr int total = stats + 2 * 8 * 32 * 8;
};
static_assert(LStatsLdsHL32::total <= 160 * 1024, "LDS layout");

__global__ __launch_bounds__(512) void retr_logit_stats_hl32_kernel(
    const _Float16* __restrict__ qh, const _Float16* __restrict__ ql,  // [T, 256, 256]
    const float* __restrict__ cy, const float* __restrict__ cx,        // [T, H, 256], [T, W, 256]
    const float* __restrict__ c3g,                                     // [T, 256]
    const _Float16* __restrict__ feat, const _Float16* __restrict__ feat_lo,
    const _Float16* __restrict__ aux,                                  // the 16-byte rows of retr_stats_hl.hip (rstd_k = bytes 8 .. 11)
    float2* __restrict__ out,                                          // [T, HW]
    int HW, int H, int W, int tiles_per_chunk) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    using Lds = LStatsLdsHL32;
    constexpr int NF = Lds::kNF, A = Lds::kA, LP = 256, TPX = 32;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
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
    // every register loaded above is consumed (and waited for) HERE: hipcc's wait-count pass does not see the asm waits below and would
    // otherwise wait for them inside the main loop, where a small vmcnt also waits for the LDS-DMA ring (retr_attn.hip)
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) asm volatile("" : "+v"(qfh[ks]), "+v"(qfl[ks]));
#pragma unroll
    for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(c3v[g]));
    wait_vm<0>();

    const u32x4 cys = ra_make_srd(cy + (size_t)t * H * LP, (uint32_t)(H * LP) * 4u);
    const u32x4 cxs = ra_make_srd(cx + (size_t)t * W * LP, (uint32_t)(W * LP) * 4u);
    const u32x4 frs = ra_make_srd((w >= 4 ? feat_lo : feat) + (size_t)t * HW * kD, (uint32_t)HW * kRowBytes);
    const u32x4 krs = ra_make_srd(aux + (size_t)t * HW * 8, (uint32_t)HW * kAuxRow);
    const u32x4 ors = ra_make_srd(out + (size_t)t * HW, (uint32_t)HW * 8u);
    auto ld16 = [](u32x4 srd, int off) {                            // asm + its own wait (no compiler-visible load in the loop)
        f32x4 v;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(off), "s"(srd) : "memory");
        return v;
    };
    f32x4 cxv[4];
    auto load_cx = [&](int strip) {
        int xx = TPX * strip + r;
        xx = xx < W ? xx : W - 1;
#pragma unroll
        for (int g = 0; g < 4; ++g) cxv[g] = ld16(cxs, (xx * LP + slot0 + 8 * g) * 4);
    };
    load_cx(strip0);

    // ---- staging: waves 0 .. 3 rows 8 w .. 8 w + 7 of the hi tile, waves 4 .. 7 the same rows of the lo tile (four pieces of two rows);
    // wave 7: the Cy row; wave 6: rstd_k of the 32 pixels
    const int rb = 8 * (w & 3);
    int voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = rb + 2 * i + h;
        voff[i] = row * kRowBytes + (((lane & 31) ^ swz(row)) * 16);
    }
    const int nb = 4 + (w >= 6 ? 1 : 0);                            // DMA instructions of one batch of this wave
    int ds = strip0, dy = row0;
    auto stage = [&](int tile) {
        if (tile >= nt) return;
        const uint32_t st = __builtin_amdgcn_readfirstlane(lds0 + Lds::ring + (tile % NF) * Lds::kStage + (w >> 2) * kTileBytes + rb * kRowBytes);
        const int px0 = dy * W + TPX * ds;
        const int soff = __builtin_amdgcn_readfirstlane(px0 * kRowBytes);
        if (px0 + TPX <= HW) {
#pragma unroll
            for (int i = 0; i < 4; ++i) ra_dma16(frs, st + i * 1024, voff[i], soff);
        } else {                                                     // last row of a ragged strip: clamp the source rows (not stored)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = rb + 2 * i + h;
                const int src = px0 + row < HW ? row : HW - 1 - px0;
                ra_dma16(frs, st + i * 1024, src * kRowBytes + (((lane & 31) ^ swz(row)) * 16), soff);
            }
        }
        if (w == 7) {
            ra_dma16_cached(cys, __builtin_amdgcn_readfirstlane(lds0 + Lds::yring + (tile % NF) * 1024), dy * LP * 4 + lane * 16);
        } else if (w == 6) {                                         // 4 B per lane: pixels px0 + (lane & 31), clamped into the frame
            int px = px0 + (lane & 31);
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
        const int xx = TPX * fs + r;
        const int pxs = fy * W + xx;
        const bool mine = have && h == 0 && (r >> 2) == w && xx < W;
        const f32x2 val = {mall, 1.f / den};
        const int so = mine ? pxs * 8 : 0x7ffffff0;                  // out of range -> dropped by the hardware range check
        asm volatile("buffer_store_dwordx2 %0, %1, %2, 0 offen" : : "v"(val), "v"(so), "s"(ors) : "memory");
        if (have) {
            ++fy;
            if (fy == H) { fy = 0; ++fs; }
        }
    };

    // ---- prologue: batches 0 .. A in flight; tile 0 landed
#pragma unroll
    for (int b = 0; b <= A; ++b) stage(b);
    {
        int younger = nt - 1;
        younger = younger < 0 ? 0 : (younger > A ? A : younger);
        wait_vm_dyn(nb * younger);
    }
    const uint32_t lane_row = lds0 + Lds::ring + r * kRowBytes + ((h ^ swz(r)) << 4);
    auto frag = [&](uint32_t tb, int ks) {
        return *reinterpret_cast<SVPS_LDS const f16x8*>((uintptr_t)((tb ^ ((ks & 7) << 5)) + 256 * (ks >> 3)));
    };
    int ts = strip0, ty = row0;
    for (int it = 0; it <= nt; ++it) {
        wg_barrier();                                                // B(it): tile it has landed for every wave; the statistics of tile it-1 are in LDS
        combine(it - 1);
        if (it == nt) break;
        const uint32_t tb = lane_row + (uint32_t)(it % NF) * Lds::kStage;
        f16x8 kf[2][4];
        auto load_grp = [&](int buf, int g) {
            kf[buf][0] = frag(tb, g);
            kf[buf][1] = frag(tb, g + 8);
            kf[buf][2] = frag(tb + kTileBytes, g);
            kf[buf][3] = frag(tb + kTileBytes, g + 8);
        };
        load_grp(0, 0);
        // Cy row and rstd_k of the tile (staged with it)
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
        const float rk = *reinterpret_cast<const float*>(smem + Lds::kring + (it % NF) * 256 + r * 4) * kLog2e;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (g < 7) load_grp((g + 1) & 1, g + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ks = g + 8 * u;
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qfl[ks], kf[g & 1][u], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qfh[ks], kf[g & 1][2 + u], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qfh[ks], kf[g & 1][u], s, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        stage(it + A + 1);                                           // its stage held tile it-1, which every wave left before B(it)
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
        // tile it+1: this wave's pieces landed. Younger operations than its batch: the batches it+2 .. it+A+1 and the stores of combine()
        // of the iterations in between
        if (it >= A + 1 && it + A + 1 < nt) wait_vm_dyn(A * (nb + 1));
        else wait_vm<0>();
        ++ty;
        if (ty == H) { ty = 0; ++ts; if (it + 1 < nt) load_cx(ts); }
    }
}

int retr_logit_stats_hl32_launch(const void* qh, const void* ql, const float* cy, const float* cx, const float* c3, const void* feat_hi,
                                 const void* feat_lo, const void* aux, void* stats, int T, int H, int W, int chunks, int tiles_per_chunk,
                                 void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    auto kern = retr_logit_stats_hl32_kernel;
    static SvpsLdsAttr attr;
    if (hipError_t ae = attr.ensure(reinterpret_cast<const void*>(kern), LStatsLdsHL32::total); ae != hipSuccess) return (int)ae;
    hipLaunchKernelGGL(kern, dim3(chunks, T), dim3(512), LStatsLdsHL32::total, stream, static_cast<const _Float16*>(qh),
                       static_cast<const _Float16*>(ql), cy, cx, c3, static_cast<const _Float16*>(feat_hi), static_cast<const _Float16*>(feat_lo),
                       static_cast<const _Float16*>(aux), static_cast<float2*>(stats), H * W, H, W, tiles_per_chunk);
    return (int)hipGetLastError();
}

int retr_attn_hl32_launch(const void* qh, const void* ql, const float* cy, const float* cx, const float* c3, const void* feat_hi,
                          const void* feat_lo, const void* aux, float* partial, int T, int L, int H, int W, int chunks, int tiles_per_chunk,
                          int LP, int Lrow, int slot_off, const void* ext_stats, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    using Lds = RetrLdsHL32;
    const _Float16* qh_ = static_cast<const _Float16*>(qh);
    const _Float16* ql_ = static_cast<const _Float16*>(ql);
    const _Float16* fh_ = static_cast<const _Float16*>(feat_hi);
    const _Float16* fl_ = static_cast<const _Float16*>(feat_lo);
    const _Float16* a_ = static_cast<const _Float16*>(aux);
    const int HW = H * W;
    if (ext_stats) {
        auto kern = retr_attn_hl32_kernel<0, true>;
        static SvpsLdsAttr attr;
        if (hipError_t ae = attr.ensure(reinterpret_cast<const void*>(kern), Lds::total); ae != hipSuccess) return (int)ae;
        hipLaunchKernelGGL(kern, dim3(chunks, T), dim3(512), Lds::total, stream, qh_, ql_, cy, cx, c3, fh_, fl_, a_, partial, L, HW, H, W,
                           tiles_per_chunk, LP, Lrow, slot_off, static_cast<const float2*>(ext_stats));
        return (int)hipGetLastError();
    }
    auto kern = retr_attn_hl32_kernel<0, false>;
    int slot = 0;
#ifdef SVPS_RETR_ABLATE
    // diagnostic build only (tools/ablate.sh builds it as a separate library): timing-only variants that return wrong results
    static const int ablate = [] { const char* a = getenv("SVPS_RETR_ABLATE"); return a ? atoi(a) : 0; }();
    if (ablate == 1) { kern = retr_attn_hl32_kernel<1, false>; slot = 1; }
    else if (ablate == 2) { kern = retr_attn_hl32_kernel<2, false>; slot = 2; }
    else if (ablate == 4) { kern = retr_attn_hl32_kernel<4, false>; slot = 3; }
    else if (ablate == 8) { kern = retr_attn_hl32_kernel<8, false>; slot = 4; }
    else if (ablate == 16) { kern = retr_attn_hl32_kernel<16, false>; slot = 5; }
#endif
    static SvpsLdsAttr attrs[6];
    if (hipError_t ae = attrs[slot].ensure(reinterpret_cast<const void*>(kern), Lds::total); ae != hipSuccess) return (int)ae;
    hipLaunchKernelGGL(kern, dim3(chunks, T), dim3(512), Lds::total, stream, qh_, ql_, cy, cx, c3, fh_, fl_, a_, partial, L, HW, H, W,
                       tiles_per_chunk, LP, Lrow, slot_off, (const float2*)nullptr);
    return (int)hipGetLastError();
}

}  // namespace svps

#ifdef SVPS_RETR_STAMP
extern "C" int svps_retr32_debug_read(unsigned long long* stamps) {
    return (int)hipMemcpyFromSymbol(stamps, HIP_SYMBOL(svps::retr32_stamps), sizeof(unsigned long long) * 2 * 8 * 8);
}
#endif
