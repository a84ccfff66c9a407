// K3'' - per-pixel LayerNorm statistics of the key / value projections for ALL retriever stages of a pyramid level in ONE read of
// the fused map: four waves, one per SIMD, 512 registers each (gfx950).
//
// MaskDynamicConv.forward (mmdet/models/detectors/dynamic_mask_head.py:428-433); the two (or one) stages of a level
// (MultiScaleDynamicMaskHead.forward :190-215) read the same map f with their own to_k / to_v / norm_k / norm_v. The algebra,
// operands and output format are those of retr_stats.hip (K3'): per stage s
//     rstd_k(p) = 1 / sqrt(|R_k,s f_p + Ty_s[y] + Tx_s[x] + r_k,s|^2 / 256 + eps)      rstd_v(p) = 1 / sqrt(|R_v,s f_p + r_v,s|^2 / 256 + eps)
// with [W~ | b~] = Q [R | r] (R upper triangular, fp16; 36 of 64 non-zero 32 x 32 blocks), written as ONE 16-byte aux row per
// pixel and stage: { 1, hi(sigma_v), lo(sigma_v), 0 } fp16, { rstd_k, rstd_v } fp32 - what the retriever stages with every pixel.
//
// Why a second form. K3' holds one stage's factors in 8 waves x 72 registers and streams the map once PER STAGE (SURVEY 8 f2(i) asks
// for one pass per level). Two stages' factors are 288 registers per lane of one SIMD: they only fit when a wave owns its SIMD's
// whole 512-entry file (256 of them AGPRs, which an MFMA takes as its A operand directly). The kernel is matrix-bound by design:
// 36 MFMA 32x32x16 per tile, SIMD and stage against ~5 vector instructions per MFMA, the regime in which a lone in-order wave
// (one instruction per ~4 cycles) keeps the matrix pipe fed.
//
// Wave sb owns row blocks (sb, 7 - sb) of R_k and R_v of every stage (NK0 = 2 (8 - sb) and NK1 = 2 (sb + 1) k-steps: 18 fragments
// per factor), the Tx + r_k rows of its lanes' pixel column (registers: the initial value of the key accumulators; a workgroup
// never leaves its 32-pixel column strip), and reads r_v and the tile's Ty row from LDS. Per tile and stage four chains
// (key / value x row block); a fragment read feeds the key and the value MFMA of its k-step. Phases (row block rb0: stage 0,
// stage 1; row block rb1: stage 0, stage 1) alternate between two accumulator pairs, and the sum of squares of one phase runs in
// the shadow of the next. The per-wave sums cross the four waves through LDS; wave s finishes stage s two tiles later (rsqrt,
// sigma_v hi / lo) and stores the tile's 32 aux rows with one instruction (512 contiguous bytes, whole memory lines).
// Branch-free loop, constant vmcnt waits, 6-deep ring of 16-KiB tiles (LDS-DMA, swizzled on the source side; each wave converts
// the four 1-KiB pieces it requested bf16 -> fp16 in place): see retr_attn4.hip for the conventions shared with the retriever.
#include <type_traits>

#include "common.h"
#include "../../include/slotvps_hip.h"

namespace svps {
namespace s4 {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) __fp16 fp16x2_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

constexpr int kFN = 6;                 // ring depth: tiles it (statistics), it+1 (fp16), it+2 (converting), it+3 .. it+4 (in flight) + the request of the iteration
constexpr int kA = kFN - 1;            // batch it + kA is requested in iteration it
constexpr int kYN = kFN + 1;           // Ty-row ring: one deeper than the tile ring - the row of tile it-1 is still read in iteration it (sums of
                                       // squares of its last phase) when the row of tile it+kA is requested
constexpr int kFB = 5;                 // row-fragment ring: kFB - 1 fragments ahead
constexpr int kMaxStages = 2;
constexpr int kTxRow = 1040;           // bytes per pixel row of the LDS Tx table (256 floats + 16: conflict-free 16-byte reads across pixel rows)

template <int NS>
struct Lds {
    static constexpr int fring = 0;                              // kFN x 16 KiB (tile bases are multiples of 512 B: fragment address XORs)
    static constexpr int yring = fring + kFN * kTileBytes;       // kYN x NS x 1 KiB: the tile's Ty row of every stage
    static constexpr int x1 = yring + kYN * NS * 1024;           // [3 tiles][NS][2 proj][32 px][4 waves] fp32 sums of squares
    static constexpr int rbv = x1 + 3 * NS * 1024;               // [NS][256] fp32 r_v
    static constexpr int txt = rbv + NS * 1024;                  // stage 1 only: [32 px][256] fp32 Tx + r_k (stage 0 keeps its rows in registers)
    static constexpr int total = txt + (NS > 1 ? 32 * 1040 : 0);
};

#define S4_FENCE() __builtin_amdgcn_sched_barrier(0)

#ifdef S4_STAMP
// diagnostic build only (make stamp4, tools/retr4_stamps.py --stats): s_memtime stamps of the four waves of ONE workgroup, iterations
// 8 .. 15, kept in LDS behind the kernel's own data and copied out at the end; (s_memtime, s_memrealtime) around every workgroup's loop
__device__ unsigned long long s4_stamps[4][8][8];
__device__ unsigned long long s4_clock[4096][4];
#define S4_STAMP_AT(pt)                                                                                          \
    do {                                                                                                         \
        S4_FENCE();                                                                                              \
        if (stamp_wg && it >= 8 && it < 16) {                                                                    \
            unsigned long long t_;                                                                               \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                           \
            if (lane == 0) reinterpret_cast<unsigned long long*>(smem + L::total)[(sb * 8 + (it - 8)) * 8 + (pt)] = t_; \
        }                                                                                                        \
        S4_FENCE();                                                                                              \
    } while (0)
#else
#define S4_STAMP_AT(pt) do {} while (0)
#endif

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}

// NS MFMA slots with the vector steps [i * NSTEP / NS, (i + 1) * NSTEP / NS) behind slot i
template <int NSL, int NSTEP, class M, class St>
__device__ __forceinline__ void phase(M&& mfma, St&& step) {
    sfor<0, NSL>([&](auto I) {
        constexpr int i = decltype(I)::value;
        mfma(I);
        S4_FENCE();
        sfor<i * NSTEP / NSL, (i + 1) * NSTEP / NSL>(step);
        S4_FENCE();
    });
}

// v_mfma_f32_32x32x16_f16 with explicit register classes (hipcc pads no hazard around an asm statement: a reader of an accumulator
// other than the next MFMA of its chain first passes settle(): 19 wait states). The accumulators live in VGPRs (the vector ALU reads
// them); the A operand - a resident fragment of R - in an AGPR or a VGPR.
template <bool AG>
__device__ __forceinline__ void mfma_acc(f32x16& acc, const f16x8& a, const f16x8& b) {
    if constexpr (AG) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// first MFMA of a key chain: D = A B + C with C = the resident Tx + r_k rows (no copy of the 16 registers)
template <bool AG>
__device__ __forceinline__ void mfma_init(f32x16& acc, const f16x8& a, const f16x8& b, const f32x16& c) {
    if constexpr (AG) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(acc) : "a"(a), "v"(b), "v"(c));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(acc) : "v"(a), "v"(b), "v"(c));
}
__device__ __forceinline__ void settle2(f32x16& x, f32x16& y) { asm volatile("s_nop 15\n\ts_nop 2" : "+v"(x), "+v"(y)); }

__device__ __forceinline__ void dma16_nt(u32x4 srd, uint32_t lds_addr, int voff, int soff) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %2, %3, %4 offen nt lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_addr), "v"(voff), "s"(srd), "s"(soff)
        : "memory");
}
__device__ __forceinline__ void dma16(u32x4 srd, uint32_t lds_addr, int voff, int soff) {
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
__device__ __forceinline__ float half_swap_sum(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ f32x4 lds4(uint32_t a) { return *reinterpret_cast<SVPS_LDS const f32x4*>((uintptr_t)a); }
__device__ __forceinline__ f16x8 lds8h(uint32_t a) { return *reinterpret_cast<SVPS_LDS const f16x8*>((uintptr_t)a); }

struct StageArgs {
    const float* ty;         // [H, 256]  R_k[:, :128] ytab[y]
    const float* tx;         // [W, 256]  R_k[:, 128:] xtab[x]
    const _Float16* rk;      // [256, 256] fp16 upper triangular
    const _Float16* rv;
    const float* rbk;        // [256]
    const float* rbv;
    __bf16* aux;             // [T, HW, 8] (16-bit words): one 16-byte row per pixel
    float eps_k, eps_v;
};
struct Args {
    const __bf16* feat;      // [T, HW, 256]
    StageArgs st[kMaxStages];
    int HW, H, W, tiles_per_chunk, chunks_per_strip;
};

// fragment `idx` of factor (stage s, projection proj, row-block slot j): does it live in an AGPR? The first 64 fragments in the order
// stage 0 (k, v), stage 1 (v, then k) do (256 registers); the remaining fragments of stage 1's key factor stay in VGPRs.
template <int NS, int SB>
__device__ __forceinline__ constexpr bool in_agpr(int s, int proj, int j, int idx) {
    constexpr int NK0 = 2 * (8 - SB), NK1 = 2 * (SB + 1);
    const int within = j == 0 ? idx : NK0 + idx;                      // 0 .. 17 inside the factor
    int ord = 0;
    if (s == 0) ord = (proj == 0 ? 0 : 18) + within;
    else ord = 36 + (proj == 1 ? 0 : 18) + within;
    (void)NK1;
    return ord < 64;
}

template <int NS, int SB>
__device__ __forceinline__ void role(const Args& A) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    using L = Lds<NS>;
    constexpr int sb = SB;
    constexpr int rb0 = SB, rb1 = 7 - SB;
    constexpr int NK0 = 2 * (8 - SB), NK1 = 2 * (SB + 1);            // k-steps of the two row blocks (18 fragments per factor)
    const int lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const int t = blockIdx.y, c = blockIdx.x;
    const int H = A.H, W = A.W, HW = A.HW;
    const int strip = c / A.chunks_per_strip;
    const int y0 = (c - strip * A.chunks_per_strip) * A.tiles_per_chunk;
    int nt = H - y0;
    nt = nt < A.tiles_per_chunk ? nt : A.tiles_per_chunk;           // >= 1 by construction of the grid
    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>((SVPS_LDS const void*)smem);
#ifdef S4_STAMP
    const bool stamp_wg = blockIdx.x == 3 && blockIdx.y == 2;
#endif
    const int x0 = kTilePx * strip;
    const bool live = x0 + r < W;                                   // pixels past the right edge of the map: not stored

    // ---- r_v of every stage -> LDS ----------------------------------------------------------------------------------------
#pragma unroll
    for (int s = 0; s < NS; ++s) reinterpret_cast<float*>(smem + L::rbv)[s * 256 + threadIdx.x] = A.st[s].rbv[threadIdx.x];
    // ---- Tx + r_k of this lane's pixel column and accumulator rows (acc register 4 g + i <-> row 32 rb + 8 g + 4 h + i) -----
    // (stage 0: registers - the C operand of the first MFMA of a key chain; stage 1: an LDS table, read into the accumulator)
    f32x16 txr[1][2];
    if constexpr (NS > 1) {
        const int tid = threadIdx.x;
        const float rb = A.st[1].rbk[tid];
        float* txl = reinterpret_cast<float*>(smem + L::txt);
#pragma unroll 4
        for (int px = 0; px < 32; ++px) {
            int xx = x0 + px;
            xx = xx < W ? xx : W - 1;
            txl[px * (kTxRow / 4) + tid] = A.st[1].tx[(size_t)xx * kD + tid] + rb;
        }
    }
    {
        int xx = x0 + r;
        xx = xx < W ? xx : W - 1;
#pragma unroll
        for (int s = 0; s < 1; ++s) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int rb = j ? rb1 : rb0;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = 32 * rb + 8 * g + 4 * h;
                    const f32x4 a = *reinterpret_cast<const f32x4*>(A.st[s].tx + (size_t)xx * kD + row);
                    const f32x4 b = *reinterpret_cast<const f32x4*>(A.st[s].rbk + row);
#pragma unroll
                    for (int i = 0; i < 4; ++i) txr[s][j][4 * g + i] = a[i] + b[i];
                }
                asm volatile("" : "+v"(txr[s][j]));
            }
        }
    }
    S4_FENCE();
    // ---- resident factors, loaded and pinned to their register class in small groups (every load waited for HERE: hipcc's wait-count
    // pass does not see the asm waits of the main loop and would otherwise drain the LDS-DMA ring inside it) ---------------------------
    f16x8 wk0[NS][NK0], wk1[NS][NK1], wv0[NS][NK0], wv1[NS][NK1];
    {
        const size_t w0off = (size_t)(32 * rb0 + r) * kD + 32 * rb0 + 8 * h, w1off = (size_t)(32 * rb1 + r) * kD + 32 * rb1 + 8 * h;
        sfor<0, NS>([&](auto S_) {
            constexpr int s = decltype(S_)::value;
            sfor<0, NK0>([&](auto I) {
                constexpr int i = decltype(I)::value;
                wk0[s][i] = *reinterpret_cast<const f16x8*>(A.st[s].rk + w0off + 16 * i);
                wv0[s][i] = *reinterpret_cast<const f16x8*>(A.st[s].rv + w0off + 16 * i);
                if constexpr (in_agpr<NS, SB>(s, 0, 0, i)) asm volatile("" : "+a"(wk0[s][i])); else asm volatile("" : "+v"(wk0[s][i]));
                if constexpr (in_agpr<NS, SB>(s, 1, 0, i)) asm volatile("" : "+a"(wv0[s][i])); else asm volatile("" : "+v"(wv0[s][i]));
                if constexpr ((i & 1) == 1) S4_FENCE();
            });
            S4_FENCE();
            sfor<0, NK1>([&](auto I) {
                constexpr int i = decltype(I)::value;
                wk1[s][i] = *reinterpret_cast<const f16x8*>(A.st[s].rk + w1off + 16 * i);
                wv1[s][i] = *reinterpret_cast<const f16x8*>(A.st[s].rv + w1off + 16 * i);
                if constexpr (in_agpr<NS, SB>(s, 0, 1, i)) asm volatile("" : "+a"(wk1[s][i])); else asm volatile("" : "+v"(wk1[s][i]));
                if constexpr (in_agpr<NS, SB>(s, 1, 1, i)) asm volatile("" : "+a"(wv1[s][i])); else asm volatile("" : "+v"(wv1[s][i]));
                if constexpr ((i & 1) == 1) S4_FENCE();
            });
            S4_FENCE();
        });
    }
    wait_vm<0>();

    // ---- LDS-DMA (branch-free: a batch past the end of the chunk goes through a descriptor of ZERO records) -----------------------
    // wave sb stages rows 8 sb .. 8 sb + 7 of every tile (4 pieces); wave 2 + s the Ty row of stage s
    constexpr bool kTyWave = SB >= 2 && SB - 2 < NS;
    constexpr int nb = 4 + (kTyWave ? 1 : 0);                        // DMA instructions of one batch of this wave
    constexpr int nst = SB < NS ? 1 : 0;                             // aux stores per iteration of this wave (they count in vmcnt too)
    const uint64_t fbase = reinterpret_cast<uint64_t>(A.feat + (size_t)t * HW * kD);
    const uint64_t ybase = reinterpret_cast<uint64_t>(A.st[kTyWave ? SB - 2 : 0].ty);
    const uint64_t abase = reinterpret_cast<uint64_t>(A.st[SB < NS ? SB : 0].aux + (size_t)t * HW * 8);
    const uint32_t fs0 = __builtin_amdgcn_readfirstlane((uint32_t)fbase), fs1 = __builtin_amdgcn_readfirstlane((uint32_t)(fbase >> 32) & 0xffffu);
    const uint32_t ys0 = __builtin_amdgcn_readfirstlane((uint32_t)ybase), ys1 = __builtin_amdgcn_readfirstlane((uint32_t)(ybase >> 32) & 0xffffu);
    const uint32_t as0 = __builtin_amdgcn_readfirstlane((uint32_t)abase), as1 = __builtin_amdgcn_readfirstlane((uint32_t)(abase >> 32) & 0xffffu);
    const uint32_t frec = (uint32_t)HW * kRowBytes, yrec = (uint32_t)H * 1024u;
    const u32x4 asrd = {as0, as1, (uint32_t)HW * 16u, 0x00020000u};
    int voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * sb + 2 * i + h;
        voff[i] = row * kRowBytes + (((lane & 31) ^ swz(row)) * 16);
    }
    auto dma_piece = [&](int i, uint32_t off_d, uint32_t yoff_d, int px0_d, int yrow_d, bool ok) {
        if (i < 4) {
            const u32x4 srd = {fs0, fs1, ok ? frec : 0u, 0x00020000u};
            dma16_nt(srd, lds0 + L::fring + off_d + sb * 4096 + i * 1024, voff[i], px0_d * kRowBytes);
        } else if (kTyWave) {
            const u32x4 srd = {ys0, ys1, ok ? yrec : 0u, 0x00020000u};
            dma16(srd, lds0 + L::yring + yoff_d + (SB - 2) * 1024, lane * 16, yrow_d * 1024);
        }
    };
    auto ring_next = [](uint32_t off) { return off + kTileBytes == (uint32_t)kFN * kTileBytes ? 0u : off + kTileBytes; };
    const uint32_t cv_lane = lds0 + L::fring + sb * 4096 + lane * 16;
    u32x4 cvw[2];
    auto convert_load = [&](uint32_t off, int i) { cvw[i & 1] = *reinterpret_cast<SVPS_LDS const u32x4*>((uintptr_t)(cv_lane + off + i * 1024)); };
    auto convert_store = [&](uint32_t off, int i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const fp16x2_t pk = __builtin_amdgcn_cvt_pkrtz(__uint_as_float(cvw[i & 1][k] << 16), __uint_as_float(cvw[i & 1][k] & 0xffff0000u));
            cvw[i & 1][k] = __builtin_bit_cast(uint32_t, pk);
        }
        *reinterpret_cast<SVPS_LDS u32x4*>((uintptr_t)(cv_lane + off + i * 1024)) = cvw[i & 1];
    };

    const uint32_t lane_row = lds0 + L::fring + r * kRowBytes + ((h ^ swz(r)) << 4);
    auto frag = [&](uint32_t tb, int ks) { return lds8h((tb ^ ((ks & 7) << 5)) + 256 * (ks >> 3)); };
    const uint32_t x1_w = lds0 + L::x1 + r * 16 + sb * 4;            // this wave's word of pixel r: + tile slot * NS KiB + stage KiB + proj * 512
    const uint32_t x1_r = lds0 + L::x1 + r * 16;

    // ---- state that crosses iterations ------------------------------------------------------------------------------------
    f32x16 ak[2], av[2];                        // two accumulator pairs: phase p uses pair p & 1
    float sqk[NS], sqv[NS];                     // running sums of squares of a tile (row block rb0, then + rb1) per stage
#pragma unroll
    for (int s = 0; s < NS; ++s) { sqk[s] = 0.f; sqv[s] = 0.f; }
#pragma unroll
    for (int i = 0; i < 16; ++i) { ak[1][i] = 0.f; av[1][i] = 0.f; }
    f16x8 fb[kFB];
    const float eps_k = A.st[SB < NS ? SB : 0].eps_k, eps_v = A.st[SB < NS ? SB : 0].eps_v;

    // One iteration (it = 0 .. nt + 1): the four (two) chains of every stage on tile it, the sums of squares of the previous phase in
    // the shadow of each phase, the finish + aux store of tile it-2 by wave s for stage s, the conversion of tile it+2 and the request
    // of tile it+kA. Iterations past the chunk run on stale tiles; their sums are never stored.
    constexpr int NPH = 2 * NS;                                      // phases of a tile: (rb0, stage 0), (rb0, stage 1), (rb1, stage 0), (rb1, stage 1)
    auto body = [&](int it, uint32_t off_l, uint32_t yoff_l, uint32_t off_c, uint32_t off_d, uint32_t yoff_d, int px0_d, int yrow_d,
                    uint32_t x1w_cur, uint32_t x1w_prev, uint32_t x1r_fin) {
        const uint32_t tb_l = lane_row + off_l;
        const bool dma_ok = it + kA < nt;
        // the row fragments of the tile's phases form ONE stream (global index G = fragments of the earlier phases + f, in
        // fb[G % kFB]); the odd slot of fragment G requests fragment G + kFB - 1, which may belong to the next phase
        auto load_stream = [&](auto G_) {
            constexpr int G = decltype(G_)::value;
            if constexpr (G < NS * (NK0 + NK1)) {
                constexpr bool second = G >= NS * NK0;
                constexpr int f = second ? (G - NS * NK0) % NK1 : G % NK0;
                fb[G % kFB] = frag(tb_l, 2 * (second ? rb1 : rb0) + f);
            }
        };
        // finish of tile it-2: the pixel this lane stores
        const int fy = y0 + it - 2;
        const bool fin_ok = it >= 2 && it - 2 < nt && live && h == 0;
        const int aoff = fin_ok ? (fy * W + x0 + r) * 16 : 0x7ffffff0;       // out of range -> dropped by the hardware range check
        f32x4 fk, fv;
        if constexpr (SB < NS) {
            fk = lds4(x1_r + x1r_fin + SB * 1024);
            fv = lds4(x1_r + x1r_fin + SB * 1024 + 512);
        }

        S4_STAMP_AT(0);
        sfor<0, NPH>([&](auto P) {
            constexpr int p = decltype(P)::value;
            constexpr int j = p / NS, s = p % NS;                    // row-block slot and stage of this phase
            constexpr int rb = j ? rb1 : rb0, NK = j ? NK1 : NK0, pr = p & 1;
            // the previous phase (of this tile, or the last one of the previous tile): its sums of squares run in this phase's shadow
            constexpr int q = (p + NPH - 1) % NPH, qj = q / NS, qs = q % NS, qrb = qj ? rb1 : rb0, qr = q & 1;
            // value accumulator: starts from r_v (LDS broadcast read); key accumulator: the first MFMA takes Tx + r_k as its C operand
            {
                const uint32_t rva = lds0 + L::rbv + s * 1024 + (32 * rb + 4 * h) * 4;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v4 = lds4(rva + 32 * g);
#pragma unroll
                    for (int i = 0; i < 4; ++i) av[pr][4 * g + i] = v4[i];
                }
            }
            if constexpr (s > 0) {
                const uint32_t txa = lds0 + L::txt + r * kTxRow + (32 * rb + 4 * h) * 4;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v4 = lds4(txa + 32 * g);
#pragma unroll
                    for (int i = 0; i < 4; ++i) ak[pr][4 * g + i] = v4[i];
                }
            }
            constexpr int base = p < NS ? p * NK0 : NS * NK0 + (p - NS) * NK1;       // global index of this phase's fragment 0
            f32x4 tyv[2];
            float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
            const uint32_t ty_prev = lds0 + L::yring + (q < p ? yoff_l : (yoff_l == 0 ? (kYN - 1) * NS * 1024u : yoff_l - NS * 1024u)) + qs * 1024 +
                                     (32 * qrb + 4 * h) * 4;
            auto step = [&](auto K) {
                constexpr int k = decltype(K)::value;
                // ---- sums of squares of phase q (pair qr): key rows + Ty, value rows
                if constexpr (k == 0) {
                    settle2(ak[qr], av[qr]);
                    tyv[0] = lds4(ty_prev);
                    tyv[1] = lds4(ty_prev + 32);
                }
                if constexpr (k >= 1 && k < 5) {
                    constexpr int g = k - 1;
                    const f32x4 tg = tyv[g & 1];
                    if constexpr (g + 2 < 4) tyv[g & 1] = lds4(ty_prev + 32 * (g + 2));
                    const float u0 = ak[qr][4 * g] + tg[0], u1 = ak[qr][4 * g + 1] + tg[1];
                    const float u2 = ak[qr][4 * g + 2] + tg[2], u3 = ak[qr][4 * g + 3] + tg[3];
                    q0 = fmaf(u0, u0, q0); q1 = fmaf(u1, u1, q1); q0 = fmaf(u2, u2, q0); q1 = fmaf(u3, u3, q1);
                }
                if constexpr (k >= 5 && k < 9) {
                    constexpr int g = k - 5;
                    q2 = fmaf(av[qr][4 * g], av[qr][4 * g], q2); q3 = fmaf(av[qr][4 * g + 1], av[qr][4 * g + 1], q3);
                    q2 = fmaf(av[qr][4 * g + 2], av[qr][4 * g + 2], q2); q3 = fmaf(av[qr][4 * g + 3], av[qr][4 * g + 3], q3);
                }
                if constexpr (k == 9) {
                    if constexpr (qj == 0) {                          // first row block of its tile: start the tile's sums
                        sqk[qs] = q0 + q1;
                        sqv[qs] = q2 + q3;
                    } else {                                          // second row block: the wave's 64 rows are complete -> LDS
                        const float totk = half_swap_sum(sqk[qs] + (q0 + q1)), totv = half_swap_sum(sqv[qs] + (q2 + q3));
                        const uint32_t xw = x1_w + (q < p ? x1w_cur : x1w_prev) + qs * 1024;
                        if (h == 0) {
                            *reinterpret_cast<SVPS_LDS float*>((uintptr_t)xw) = totk;
                            *reinterpret_cast<SVPS_LDS float*>((uintptr_t)(xw + 512)) = totv;
                        }
                    }
                }
                // ---- phase 0: finish + aux store of tile it-2 (wave s: stage s); always issued (vmcnt counts on it)
                if constexpr (p == 0 && k == 10 && SB < NS) {
                    const float totk = (fk[0] + fk[1]) + (fk[2] + fk[3]), totv = (fv[0] + fv[1]) + (fv[2] + fv[3]);
                    const float vark = totk * (1.f / kD) + eps_k, varv = totv * (1.f / kD) + eps_v;
                    const float rstdk = __builtin_amdgcn_rsqf(vark), rstdv = __builtin_amdgcn_rsqf(varv);
                    const float sigma = varv * rstdv;
                    const _Float16 sh = (_Float16)sigma;
                    const _Float16 sl = (_Float16)(sigma - (float)sh);
                    const _Float16 one = (_Float16)1.0f;
                    const uint32_t w0 = (uint32_t)__builtin_bit_cast(uint16_t, one) | ((uint32_t)__builtin_bit_cast(uint16_t, sh) << 16);
                    const uint32_t w1 = (uint32_t)__builtin_bit_cast(uint16_t, sl);
                    const u32x4 row16 = {w0, w1, __float_as_uint(rstdk), __float_as_uint(rstdv)};
                    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" : : "v"(row16), "v"(aoff), "s"(asrd) : "memory");
                }
                // ---- phase 1 (or 0 when there is only one stage ... ): conversion of this wave's pieces of tile it+2
                if constexpr (p == (NPH > 2 ? 1 : 0) && k == 11) {
                    // landed: batch it+2 = everything but the batches it+3 .. it+kA-1 and the aux stores issued since its request
                    wait_vm<nb * (kA - 3) + nst * (kA - 2)>();
                    convert_load(off_c, 0);
                    convert_load(off_c, 1);
                }
                if constexpr (p == (NPH > 2 ? 2 : 1) && k >= 10 && k < 14) {
                    convert_store(off_c, k - 10);
                    if constexpr (k - 10 + 2 < 4) convert_load(off_c, k - 10 + 2);
                }
                // ---- last phase: request of tile it+kA
                if constexpr (p == NPH - 1 && k >= 10 && k < 14) dma_piece(k - 10, off_d, yoff_d, px0_d, yrow_d, dma_ok);
                if constexpr (p == NPH - 1 && k == 14 && nb == 5) dma_piece(4, off_d, yoff_d, px0_d, yrow_d, dma_ok);
            };
            phase<2 * NK, 15>(
                [&](auto I) {
                    constexpr int i = decltype(I)::value, f = i >> 1;
                    if constexpr ((i & 1) == 0) {
                        if constexpr (j == 0) {
                            if constexpr (f == 0 && s == 0) mfma_init<in_agpr<NS, SB>(s, 0, 0, 0)>(ak[pr], wk0[s][0], fb[base % kFB], txr[0][0]);
                            else mfma_acc<in_agpr<NS, SB>(s, 0, 0, f)>(ak[pr], wk0[s][f], fb[(base + f) % kFB]);
                        } else {
                            if constexpr (f == 0 && s == 0) mfma_init<in_agpr<NS, SB>(s, 0, 1, 0)>(ak[pr], wk1[s][0], fb[base % kFB], txr[0][1]);
                            else mfma_acc<in_agpr<NS, SB>(s, 0, 1, f)>(ak[pr], wk1[s][f], fb[(base + f) % kFB]);
                        }
                    } else {
                        if constexpr (j == 0) mfma_acc<in_agpr<NS, SB>(s, 1, 0, f)>(av[pr], wv0[s][f], fb[(base + f) % kFB]);
                        else mfma_acc<in_agpr<NS, SB>(s, 1, 1, f)>(av[pr], wv1[s][f], fb[(base + f) % kFB]);
                        load_stream(std::integral_constant<int, base + f + kFB - 1>{});
                    }
                },
                step);
            S4_STAMP_AT(p + 1);
        });
    };

    // ---- prologue: batches 0 .. kA-1 requested (+ as many dummy stores as the steady state has in flight); tiles 0, 1 converted ----
    {
        uint32_t off = 0, yoff = 0;
        int px0 = y0 * W + x0, yrow = y0;
#pragma unroll
        for (int b = 0; b < kA; ++b) {
#pragma unroll
            for (int i = 0; i < 5; ++i) { if (i < nb) dma_piece(i, off, yoff, px0, yrow, b < nt); }
            off += kTileBytes;
            yoff += NS * 1024;
            px0 += W;
            ++yrow;
        }
        if constexpr (nst) {
            const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int i = 0; i < kA - 3; ++i) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" : : "v"(z), "v"(0x7ffffff0), "s"(asrd) : "memory");
        }
    }
    wait_vm<nb * (kA - 2) + nst * (kA - 3)>();                       // batches 0 and 1
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { convert_load(b * kTileBytes, i); convert_store(b * kTileBytes, i); }
    }
    wg_barrier();
#ifdef S4_STAMP
    const int wg_lin = blockIdx.y * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0 && wg_lin < 4096) {
        s4_clock[wg_lin][0] = __builtin_amdgcn_s_memtime();
        s4_clock[wg_lin][1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    {
        uint32_t off_l = 0, yoff_l = 0, off_c = 2 * kTileBytes, off_d = kA * kTileBytes, yoff_d = kA * NS * 1024;
        int px0_d = (y0 + kA) * W + x0, yrow_d = y0 + kA;
        uint32_t x1w_cur = 0, x1w_prev = 2 * NS * 1024, x1r_fin = 1 * NS * 1024;   // x1 slots of tiles it, it-1, it-2 (mod 3)
        for (int it = 0; it <= nt + 1; ++it) {
            {   // the first row fragments of tile it (phase 0: row block rb0), requested before the barrier and still in flight behind it
                const uint32_t tb = lane_row + off_l;
                S4_FENCE();
#pragma unroll
                for (int f = 0; f < kFB - 1; ++f) fb[f] = frag(tb, 2 * rb0 + f);
                S4_FENCE();
            }
            S4_STAMP_AT(NPH + 1);
#ifdef S4_STAMP
            wg_barrier();
#else
            asm volatile("s_waitcnt lgkmcnt(%0)\n\ts_barrier" ::"n"(kFB - 1) : "memory");       // B(it)
#endif
            body(it, off_l, yoff_l, off_c, off_d, yoff_d, px0_d, yrow_d, x1w_cur, x1w_prev, x1r_fin);
            off_l = ring_next(off_l);
            off_c = ring_next(off_c);
            off_d = ring_next(off_d);
            yoff_l = yoff_l + NS * 1024 == (uint32_t)kYN * NS * 1024 ? 0u : yoff_l + NS * 1024;
            yoff_d = yoff_d + NS * 1024 == (uint32_t)kYN * NS * 1024 ? 0u : yoff_d + NS * 1024;
            px0_d += W;
            ++yrow_d;
            const uint32_t tmp = x1r_fin;                            // rotate the three x1 slots: the finished one becomes the next tile's
            x1r_fin = x1w_prev;
            x1w_prev = x1w_cur;
            x1w_cur = tmp;
        }
    }
    wait_vm<0>();
#ifdef S4_STAMP
    if (threadIdx.x == 0 && wg_lin < 4096) {
        s4_clock[wg_lin][2] = __builtin_amdgcn_s_memtime();
        s4_clock[wg_lin][3] = __builtin_amdgcn_s_memrealtime();
    }
    if (stamp_wg && lane < 64) (&s4_stamps[sb][0][0])[lane] = reinterpret_cast<const unsigned long long*>(smem + L::total)[sb * 64 + lane];
#endif
}

template <int NS>
__global__ __launch_bounds__(256) void retr_stats4_kernel(const Args A) {
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    switch (w) {                 // every role runs the same sequence of workgroup barriers
        case 0: role<NS, 0>(A); break;
        case 1: role<NS, 1>(A); break;
        case 2: role<NS, 2>(A); break;
        default: role<NS, 3>(A); break;
    }
}

}  // namespace s4
}  // namespace svps

namespace {
struct PlanS4 {
    int strips, cps, tpc;
    int chunks() const { return strips * cps; }
};
// chunks never leave a strip; rounds of one workgroup per CU; every workgroup pays a prologue of about six tile times
PlanS4 plan_s4(int T, int H, int W) {
    const int strips = (W + svps::kTilePx - 1) / svps::kTilePx;
    const int cus = svps_num_cus();
    double best = -1.0;
    int cps = 1;
    for (int c = 1; c <= H; ++c) {
        const int tpc = (H + c - 1) / c;
        if (c > 1 && tpc < 8) break;
        const int cc = (H + tpc - 1) / tpc;
        const long wg = (long)T * strips * cc;
        const long rounds = (wg + cus - 1) / cus;
        const double eff = (double)T * strips * H / ((double)rounds * cus * (tpc + 6));
        if (eff > best + 1e-9) { best = eff; cps = cc; }
    }
    const int tpc = (H + cps - 1) / cps;
    cps = (H + tpc - 1) / tpc;
    return {strips, cps, tpc};
}
}  // namespace

extern "C" int svps_retr_stats_level_fwd(const void* feat, int n_stages, const float* const* ty, const float* const* tx,
                                         const void* const* rk, const float* const* rbk, const float* lnk_eps,
                                         const void* const* rv, const float* const* rbv, const float* lnv_eps,
                                         void* const* aux, int T, int H, int W, int D, void* stream_) {
    if (!feat || !ty || !tx || !rk || !rbk || !lnk_eps || !rv || !rbv || !lnv_eps || !aux) return SVPS_ERR_BAD_ARG;
    if (n_stages < 1 || n_stages > svps::s4::kMaxStages) return SVPS_ERR_BAD_SHAPE;
    if (D != svps::kD || T <= 0 || H <= 0 || W <= 0) return SVPS_ERR_BAD_SHAPE;
    if ((size_t)H * W > svps::kMaxFramePixels) return SVPS_ERR_BAD_SHAPE;
    svps::s4::Args a;
    a.feat = static_cast<const __bf16*>(feat);
    for (int s = 0; s < svps::s4::kMaxStages; ++s) {
        const int q = s < n_stages ? s : 0;
        if (!ty[q] || !tx[q] || !rk[q] || !rbk[q] || !rv[q] || !rbv[q] || !aux[q]) return SVPS_ERR_BAD_ARG;
        a.st[s].ty = ty[q]; a.st[s].tx = tx[q];
        a.st[s].rk = static_cast<const _Float16*>(rk[q]); a.st[s].rv = static_cast<const _Float16*>(rv[q]);
        a.st[s].rbk = rbk[q]; a.st[s].rbv = rbv[q];
        a.st[s].aux = static_cast<__bf16*>(aux[q]);
        a.st[s].eps_k = lnk_eps[q]; a.st[s].eps_v = lnv_eps[q];
    }
    const PlanS4 p = plan_s4(T, H, W);
    a.HW = H * W; a.H = H; a.W = W; a.tiles_per_chunk = p.tpc; a.chunks_per_strip = p.cps;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    static SvpsLdsAttr attr1, attr2;
#ifdef S4_STAMP
    constexpr int kStampBytes = 2048;
#else
    constexpr int kStampBytes = 0;
#endif
    svps_prof_mark(SVPS_KERNEL_RETR_STATS, 0, stream);
    if (n_stages == 1) {
        if (hipError_t ae = attr1.ensure(reinterpret_cast<const void*>(svps::s4::retr_stats4_kernel<1>), svps::s4::Lds<1>::total + kStampBytes); ae != hipSuccess) return (int)ae;
        hipLaunchKernelGGL(svps::s4::retr_stats4_kernel<1>, dim3(p.chunks(), T), dim3(256), svps::s4::Lds<1>::total + kStampBytes, stream, a);
    } else {
        if (hipError_t ae = attr2.ensure(reinterpret_cast<const void*>(svps::s4::retr_stats4_kernel<2>), svps::s4::Lds<2>::total + kStampBytes); ae != hipSuccess) return (int)ae;
        hipLaunchKernelGGL(svps::s4::retr_stats4_kernel<2>, dim3(p.chunks(), T), dim3(256), svps::s4::Lds<2>::total + kStampBytes, stream, a);
    }
    svps_prof_mark(SVPS_KERNEL_RETR_STATS, 1, stream);
    return (int)hipGetLastError();
}

#ifdef S4_STAMP
extern "C" int svps_stats4_debug_read(unsigned long long* stamps, unsigned long long* clock) {
    hipError_t e = hipMemcpyFromSymbol(stamps, HIP_SYMBOL(svps::s4::s4_stamps), sizeof(unsigned long long) * 4 * 8 * 8);
    if (e != hipSuccess) return (int)e;
    return (int)hipMemcpyFromSymbol(clock, HIP_SYMBOL(svps::s4::s4_clock), sizeof(unsigned long long) * 4096 * 4);
}
#endif
