// K3'' - per-pixel LayerNorm statistics of the key / value projections for BOTH retriever stages of a pyramid level in ONE read of
// the fused map (gfx950).
//
// MaskDynamicConv.forward (mmdet/models/detectors/dynamic_mask_head.py:428-433); the two stages of a level
// (MultiScaleDynamicMaskHead.forward :190-215) read the same map f with their own to_k / to_v / norm_k / norm_v. The algebra,
// operands and output format are those of retr_stats.hip (K3'): per stage s
//     rstd_k(p) = 1 / sqrt(|R_k,s f_p + Ty_s[y] + Tx_s[x] + r_k,s|^2 / 256 + eps)      rstd_v(p) = 1 / sqrt(|R_v,s f_p + r_v,s|^2 / 256 + eps)
// with [W~ | b~] = Q [R | r] (R upper triangular, fp16; 36 of 64 non-zero 32 x 32 blocks), written as ONE 16-byte aux row per
// pixel and stage: { 1, hi(sigma_v), lo(sigma_v), 0 } fp16, { rstd_k, rstd_v } fp32 - what the retriever stages with every pixel.
//
// K3' streams the map once PER STAGE (SURVEY 8 f2(i) asks for one pass per level). Here 8 waves share one LDS tile ring:
// wave w = (stage w >> 2, quarter sb = w & 3) holds row blocks (sb, 7 - sb) of BOTH factors of its stage - 36 fragments = 144
// registers, kept in AGPRs, which an MFMA takes as its A operand directly - so two stages' factors (288 registers per SIMD lane)
// fit next to two waves' working sets. A SIMD hosts one wave of each stage: the matrix pipe (72 MFMA 32x32x16 per tile and
// SIMD: the bound of this kernel) is fed by whichever of the two is not in its vector phase; the stage-1 waves walk their two
// row blocks in the opposite order, so that the long chain of one wave meets the short chain + sums of squares of the other.
// (A first form with FOUR waves of 512 registers - one per SIMD, both stages per wave - measured 204 us at the finest level
// against 2 x 115 for K3': with a lone wave per SIMD every LDS round trip, DMA issue and barrier skew is exposed - 5500 cycles per
// tile against 2300 of MFMA time in its s_memtime stamps. DESIGN.md section 7.)
//
// Tile = 32 consecutive pixels of one image row; a workgroup walks DOWN a 32-pixel-wide column strip and never leaves it: the
// Tx + r_k rows of a lane's pixel column are loaded once (row block sb: 16 registers, the C operand of the chain's first MFMA;
// row block 7 - sb: an LDS table). Per tile a wave runs two phases (row block rb0, rb1), each a key and a value chain that share
// their row fragments, followed by the sums of squares; the per-wave sums cross the four waves of a stage through LDS and wave
// (s, 0) finishes stage s one tile later (rsqrt, sigma_v hi / lo) and stores the tile's 32 aux rows with one instruction (512
// contiguous bytes, whole memory lines). Branch-free loop with constant vmcnt waits: a batch past the end of the chunk is
// requested through a descriptor of zero records; 6-deep ring of 16-KiB tiles (LDS-DMA, swizzled on the source side; each wave
// converts the two 1-KiB pieces it requested bf16 -> fp16 in place).
#include <type_traits>

#include "common.h"
#include "../../include/slotvps_hip.h"

extern "C" int svps_retr_stats_fwd(const void* feat, const float* ty, const float* tx, const void* rk, const float* rbk,
                                   float lnk_eps, const void* rv, const float* rbv, float lnv_eps,
                                   void* aux, int T, int H, int W, int D, int flags, void* stream_);

namespace svps {
namespace s2 {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) __fp16 fp16x2_t;

constexpr int kFN = 6;                 // ring depth: tiles it (statistics), it+1 (fp16), it+2 (converting), it+3, it+4 (in flight) + the request of the iteration
constexpr int kA = kFN - 1;            // batch it + kA is requested in iteration it
constexpr int kYN = kFN + 1;           // Ty-row ring (one deeper: simpler than proving the last reader of a row)
#ifndef S2_FB
#define S2_FB 4
#endif
#ifndef S2_STAGGER
#define S2_STAGGER 1
#endif
constexpr int kFB = S2_FB;                 // row-fragment ring: kFB - 1 fragments ahead
constexpr int kTxRow = 528;            // bytes per pixel row of the LDS Tx table (128 floats + 16: conflict-free 16-byte reads across pixel rows)

struct Lds {
    static constexpr int fring = 0;                              // kFN x 16 KiB (tile bases are multiples of 512 B: fragment address XORs)
    static constexpr int yring = fring + kFN * kTileBytes;       // kYN x 2 KiB: the tile's Ty row of stage 0 | stage 1
    static constexpr int txt = yring + kYN * 2048;               // [2 stages][32 px][128 rows 128 .. 255] fp32 Tx + r_k (row blocks 4 .. 7)
    static constexpr int x1 = txt + 2 * 32 * kTxRow;             // [2 tiles][2 stages][2 proj][32 px][4 waves] fp32 sums of squares
    static constexpr int rbv = x1 + 2 * 2 * 1024;                // [2 stages][256] fp32 r_v
    static constexpr int total = rbv + 2 * 1024;
};
static_assert(Lds::total <= 160 * 1024, "LDS layout");

#define S2_FENCE() __builtin_amdgcn_sched_barrier(0)

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}

// v_mfma_f32_32x32x16_f16 with the resident A fragment in an AGPR and the accumulator in VGPRs (the vector ALU reads it). hipcc pads no
// hazard around an asm statement: the reader of an accumulator first passes settle2() (19 wait states).
template <bool AG = true>
__device__ __forceinline__ void mfma_acc(f32x16& acc, const f16x8& a, const f16x8& b) {
    if constexpr (AG) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// first MFMA of a key chain: D = A B + C with C = the resident Tx + r_k rows (no copy of the 16 registers)
__device__ __forceinline__ void mfma_init(f32x16& acc, const f16x8& a, const f16x8& b, const f32x16& c) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(acc) : "a"(a), "v"(b), "v"(c));
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
    StageArgs st[2];
    int HW, H, W, tiles_per_chunk, chunks_per_strip;
    int map_f16;             // the map is fp16 already: no conversion in LDS
};

// 256 registers per wave = 128 AGPRs (32 of the 36 resident fragments) + 128 VGPRs: the last two k-steps of row block 7 - sb of both
// factors stay in VGPRs (an "a" operand hipcc holds in a VGPR is copied before EVERY use, with a hazard nobody pads)
template <int NK1>
__device__ __forceinline__ constexpr bool rb1_in_agpr(int idx) { return idx < NK1 - 2; }

// ST: the wave's stage (0: waves 0 .. 3, 1: waves 4 .. 7); SB: its quarter of the row blocks
template <int ST, int SB>
__device__ __forceinline__ void role(const Args& A) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    using L = Lds;
    constexpr int sb = SB, wv = 4 * ST + SB;                         // wave number in the workgroup
    constexpr int rb0 = SB, rb1 = 7 - SB;
    constexpr int NK0 = 2 * (8 - SB), NK1 = 2 * (SB + 1);            // k-steps of the two row blocks (18 fragments per factor)
    const StageArgs& S = A.st[ST];
    const int lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const int t = blockIdx.y, c = blockIdx.x;
    const int H = A.H, W = A.W, HW = A.HW;
    const int strip = c / A.chunks_per_strip;
    const int y0 = (c - strip * A.chunks_per_strip) * A.tiles_per_chunk;
    int nt = H - y0;
    nt = nt < A.tiles_per_chunk ? nt : A.tiles_per_chunk;           // >= 1 by construction of the grid
    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>((SVPS_LDS const void*)smem);
    const int x0 = kTilePx * strip;
    const bool live = x0 + r < W;                                   // pixels past the right edge of the map: not stored

    // ---- per workgroup: r_v of both stages and the Tx + r_k rows 128 .. 255 of the strip -> LDS (thread = (stage, row)) --------
    {
        const int tid = threadIdx.x, s = tid >> 8, row = tid & 255;
        reinterpret_cast<float*>(smem + L::rbv)[s * 256 + row] = A.st[s].rbv[row];
        if (row >= 128) {
            const float rb = A.st[s].rbk[row];
            float* txl = reinterpret_cast<float*>(smem + L::txt + s * 32 * kTxRow);
#pragma unroll 4
            for (int px = 0; px < 32; ++px) {
                int xx = x0 + px;
                xx = xx < W ? xx : W - 1;
                txl[px * (kTxRow / 4) + row - 128] = A.st[s].tx[(size_t)xx * kD + row] + rb;
            }
        }
    }
    // ---- Tx + r_k of this lane's pixel column, rows of block rb0 (acc register 4 g + i <-> row 32 rb + 8 g + 4 h + i): the C operand
    // of the first MFMA of that key chain
    f32x16 txr;
    {
        int xx = x0 + r;
        xx = xx < W ? xx : W - 1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = 32 * rb0 + 8 * g + 4 * h;
            const f32x4 a = *reinterpret_cast<const f32x4*>(S.tx + (size_t)xx * kD + row);
            const f32x4 b = *reinterpret_cast<const f32x4*>(S.rbk + row);
#pragma unroll
            for (int i = 0; i < 4; ++i) txr[4 * g + i] = a[i] + b[i];
        }
        asm volatile("" : "+v"(txr));
    }
    S2_FENCE();
    // ---- resident factors -> AGPRs, loaded and pinned in small groups (every load waited for HERE: hipcc's wait-count pass does not see
    // the asm waits of the main loop and would otherwise drain the LDS-DMA ring inside it) --------------------------------------------
    f16x8 wk0[NK0], wk1[NK1], wv0[NK0], wv1[NK1];
    {
        const size_t w0off = (size_t)(32 * rb0 + r) * kD + 32 * rb0 + 8 * h, w1off = (size_t)(32 * rb1 + r) * kD + 32 * rb1 + 8 * h;
        sfor<0, NK0>([&](auto I) {
            constexpr int i = decltype(I)::value;
            wk0[i] = *reinterpret_cast<const f16x8*>(S.rk + w0off + 16 * i);
            wv0[i] = *reinterpret_cast<const f16x8*>(S.rv + w0off + 16 * i);
            asm volatile("" : "+a"(wk0[i]));
            asm volatile("" : "+a"(wv0[i]));
            if constexpr ((i & 1) == 1) S2_FENCE();
        });
        S2_FENCE();
        sfor<0, NK1>([&](auto I) {
            constexpr int i = decltype(I)::value;
            wk1[i] = *reinterpret_cast<const f16x8*>(S.rk + w1off + 16 * i);
            wv1[i] = *reinterpret_cast<const f16x8*>(S.rv + w1off + 16 * i);
            if constexpr (rb1_in_agpr<NK1>(i)) { asm volatile("" : "+a"(wk1[i])); asm volatile("" : "+a"(wv1[i])); }
            else { asm volatile("" : "+v"(wk1[i])); asm volatile("" : "+v"(wv1[i])); }
            if constexpr ((i & 1) == 1) S2_FENCE();
        });
        S2_FENCE();
    }
    wait_vm<0>();

    // ---- LDS-DMA: wave wv stages rows 4 wv .. 4 wv + 3 of every tile (2 pieces); waves 3 / 7 the Ty row of stage 0 / 1 ------------
    constexpr bool kTyWave = SB == 3;
    constexpr int nb = 2 + (kTyWave ? 1 : 0);                        // DMA instructions of one batch of this wave
    constexpr int nst = SB == 0 ? 1 : 0;                             // aux stores per iteration of this wave (they count in vmcnt too)
    const uint64_t fbase = reinterpret_cast<uint64_t>(A.feat + (size_t)t * HW * kD);
    const uint64_t ybase = reinterpret_cast<uint64_t>(S.ty);
    const uint64_t abase = reinterpret_cast<uint64_t>(S.aux + (size_t)t * HW * 8);
    const uint32_t fs0 = __builtin_amdgcn_readfirstlane((uint32_t)fbase), fs1 = __builtin_amdgcn_readfirstlane((uint32_t)(fbase >> 32) & 0xffffu);
    const uint32_t ys0 = __builtin_amdgcn_readfirstlane((uint32_t)ybase), ys1 = __builtin_amdgcn_readfirstlane((uint32_t)(ybase >> 32) & 0xffffu);
    const uint32_t as0 = __builtin_amdgcn_readfirstlane((uint32_t)abase), as1 = __builtin_amdgcn_readfirstlane((uint32_t)(abase >> 32) & 0xffffu);
    const uint32_t frec = (uint32_t)HW * kRowBytes, yrec = (uint32_t)H * 1024u;
    const u32x4 asrd = {as0, as1, (uint32_t)HW * 16u, 0x00020000u};
    int voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 4 * wv + 2 * i + h;
        voff[i] = row * kRowBytes + (((lane & 31) ^ swz(row)) * 16);
    }
    auto dma_batch = [&](uint32_t off_d, uint32_t yoff_d, int px0_d, int yrow_d, bool ok) {
        const u32x4 srd = {fs0, fs1, ok ? frec : 0u, 0x00020000u};
        dma16_nt(srd, lds0 + L::fring + off_d + wv * 2048, voff[0], px0_d * kRowBytes);
        dma16_nt(srd, lds0 + L::fring + off_d + wv * 2048 + 1024, voff[1], px0_d * kRowBytes);
        if constexpr (kTyWave) {
            const u32x4 ysrd = {ys0, ys1, ok ? yrec : 0u, 0x00020000u};
            dma16(ysrd, lds0 + L::yring + yoff_d + ST * 1024, lane * 16, yrow_d * 1024);
        }
    };
    auto ring_next = [](uint32_t off) { return off + kTileBytes == (uint32_t)kFN * kTileBytes ? 0u : off + kTileBytes; };
    auto yring_next = [](uint32_t off) { return off + 2048 == (uint32_t)kYN * 2048 ? 0u : off + 2048; };
    const uint32_t cv_lane = lds0 + L::fring + wv * 2048 + lane * 16;
    auto convert_piece = [&](uint32_t off, u32x4& w_, int i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const fp16x2_t pk = __builtin_amdgcn_cvt_pkrtz(__uint_as_float(w_[k] << 16), __uint_as_float(w_[k] & 0xffff0000u));
            w_[k] = __builtin_bit_cast(uint32_t, pk);
        }
        *reinterpret_cast<SVPS_LDS u32x4*>((uintptr_t)(cv_lane + off + i * 1024)) = w_;
    };

    const uint32_t lane_row = lds0 + L::fring + r * kRowBytes + ((h ^ swz(r)) << 4);
    auto frag = [&](uint32_t tb, int ks) { return lds8h((tb ^ ((ks & 7) << 5)) + 256 * (ks >> 3)); };
    const uint32_t x1_w = lds0 + L::x1 + ST * 1024 + r * 16 + sb * 4;   // + tile parity * 2048 (+ 512: value)
    const uint32_t x1_r = lds0 + L::x1 + ST * 1024 + r * 16;
    const uint32_t rbv_lane = lds0 + L::rbv + ST * 1024 + 4 * h * 4, ty_lane = lds0 + L::yring + ST * 1024 + 4 * h * 4;
    const uint32_t tx_lane = lds0 + L::txt + ST * 32 * kTxRow + r * kTxRow + (32 * rb1 - 128 + 4 * h) * 4;
    const float eps_k = S.eps_k, eps_v = S.eps_v;

    f32x16 ak, av;
    f16x8 fb[kFB];
    // One phase: the key and the value chain of row block `rb` (NK k-steps from 2 rb) on the tile at `tb`, then the sums of squares
    // of the lane's 2 x 16 rows (key rows + the tile's Ty row). FIRST: its first fragments were requested before the barrier.
    auto run_phase = [&](auto RB, auto NK_, auto FIRST, auto wk, auto wv_, uint32_t tb, uint32_t yoff, float& sqk, float& sqv) {
        constexpr int rb = decltype(RB)::value, NK = decltype(NK_)::value;
        constexpr bool first = decltype(FIRST)::value;
        constexpr bool c_init = rb == rb0;                           // key accumulator from the txr registers (row block rb0) or the LDS table
        {
            const uint32_t rva = rbv_lane + 32 * rb * 4;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v4 = lds4(rva + 32 * g);
#pragma unroll
                for (int i = 0; i < 4; ++i) av[4 * g + i] = v4[i];
            }
            if constexpr (!c_init) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v4 = lds4(tx_lane + 32 * g);
#pragma unroll
                    for (int i = 0; i < 4; ++i) ak[4 * g + i] = v4[i];
                }
            }
        }
        if constexpr (!first) {
#pragma unroll
            for (int f = 0; f < kFB - 1; ++f) { if (f < NK) fb[f] = frag(tb, 2 * rb + f); }
        }
        sfor<0, NK>([&](auto F) {
            constexpr int f = decltype(F)::value;
            constexpr bool ag = c_init || rb1_in_agpr<NK1>(f);       // (row block rb0: all fragments in AGPRs)
            if constexpr (f == 0 && c_init) mfma_init(ak, wk[0], fb[0], txr);
            else mfma_acc<ag>(ak, wk[f], fb[f % kFB]);
            mfma_acc<ag>(av, wv_[f], fb[f % kFB]);
            if constexpr (f + kFB - 1 < NK) fb[(f + kFB - 1) % kFB] = frag(tb, 2 * rb + f + kFB - 1);
            S2_FENCE();
        });
        // sums of squares
        const uint32_t tya = ty_lane + yoff + 32 * rb * 4;
        f32x4 tyv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) tyv[g] = lds4(tya + 32 * g);
        settle2(ak, av);
        float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float u0 = ak[4 * g] + tyv[g][0], u1 = ak[4 * g + 1] + tyv[g][1];
            const float u2 = ak[4 * g + 2] + tyv[g][2], u3 = ak[4 * g + 3] + tyv[g][3];
            q0 = fmaf(u0, u0, q0); q1 = fmaf(u1, u1, q1); q0 = fmaf(u2, u2, q0); q1 = fmaf(u3, u3, q1);
            q2 = fmaf(av[4 * g], av[4 * g], q2); q3 = fmaf(av[4 * g + 1], av[4 * g + 1], q3);
            q2 = fmaf(av[4 * g + 2], av[4 * g + 2], q2); q3 = fmaf(av[4 * g + 3], av[4 * g + 3], q3);
        }
        sqk += q0 + q1;
        sqv += q2 + q3;
    };

    // ---- prologue: batches 0 .. kA-1 requested (+ as many dummy stores as the steady state has in flight); tiles 0, 1 converted ----
    {
        uint32_t off = 0, yoff = 0;
        int px0 = y0 * W + x0, yrow = y0;
#pragma unroll
        for (int b = 0; b < kA; ++b) {
            dma_batch(off, yoff, px0, yrow, b < nt);
            off += kTileBytes;
            yoff += 2048;
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
    if (!A.map_f16) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                u32x4 w_ = *reinterpret_cast<SVPS_LDS const u32x4*>((uintptr_t)(cv_lane + b * kTileBytes + i * 1024));
                convert_piece(b * kTileBytes, w_, i);
            }
        }
    }
    wg_barrier();

    // Iteration it = 0 .. nt: statistics of tile it (two phases; the stage-1 waves walk the row blocks in the opposite order), the
    // finish + aux store of tile it-1 by wave (stage, 0), the conversion of this wave's pieces of tile it+2, the request of tile
    // it+kA. The iteration past the chunk runs on a stale tile; its sums are never stored.
    uint32_t off_l = 0, yoff_l = 0, off_c = 2 * kTileBytes, off_d = kA * kTileBytes, yoff_d = kA * 2048;
    int px0_d = (y0 + kA) * W + x0, yrow_d = y0 + kA;
    constexpr int rbA = (ST == 0 || !S2_STAGGER) ? rb0 : rb1, NKA = (ST == 0 || !S2_STAGGER) ? NK0 : NK1;   // first phase of this wave
    for (int it = 0; it <= nt; ++it) {
        const uint32_t tb_l = lane_row + off_l;
        {   // the first row fragments of tile it, requested before the barrier and still in flight behind it (LDS operations of a
            // wave complete in order: "all but the kFB - 1 youngest" covers every LDS write of the iteration the barrier publishes)
            S2_FENCE();
#pragma unroll
            for (int f = 0; f < kFB - 1; ++f) { if (f < NKA) fb[f] = frag(tb_l, 2 * rbA + f); }
            S2_FENCE();
        }
        asm volatile("s_waitcnt lgkmcnt(%0)\n\ts_barrier" ::"n"(NKA < kFB - 1 ? NKA : kFB - 1) : "memory");       // B(it)
        const uint32_t par = (it & 1) * 2048, parp = 2048 - par;
        // finish of tile it-1: sums of the four waves of this stage (written before the barrier)
        f32x4 fk, fv;
        if constexpr (SB == 0) {
            fk = lds4(x1_r + parp);
            fv = lds4(x1_r + parp + 512);
        }
        float sqk = 0.f, sqv = 0.f;
        if constexpr (ST == 0 || !S2_STAGGER) {
            run_phase(std::integral_constant<int, rb0>{}, std::integral_constant<int, NK0>{}, std::true_type{}, wk0, wv0, tb_l, yoff_l, sqk, sqv);
            run_phase(std::integral_constant<int, rb1>{}, std::integral_constant<int, NK1>{}, std::false_type{}, wk1, wv1, tb_l, yoff_l, sqk, sqv);
        } else {
            run_phase(std::integral_constant<int, rb1>{}, std::integral_constant<int, NK1>{}, std::true_type{}, wk1, wv1, tb_l, yoff_l, sqk, sqv);
            run_phase(std::integral_constant<int, rb0>{}, std::integral_constant<int, NK0>{}, std::false_type{}, wk0, wv0, tb_l, yoff_l, sqk, sqv);
        }
        {
            const float totk = half_swap_sum(sqk), totv = half_swap_sum(sqv);
            if (h == 0) {
                *reinterpret_cast<SVPS_LDS float*>((uintptr_t)(x1_w + par)) = totk;
                *reinterpret_cast<SVPS_LDS float*>((uintptr_t)(x1_w + par + 512)) = totv;
            }
        }
        if constexpr (SB == 0) {                                     // aux rows of tile it-1; always issued (vmcnt counts on it)
            const bool fin_ok = it >= 1 && live && h == 0;
            const int aoff = fin_ok ? ((y0 + it - 1) * W + x0 + r) * 16 : 0x7ffffff0;   // out of range -> dropped by the hardware range check
            const float totk = (fk[0] + fk[1]) + (fk[2] + fk[3]), totv = (fv[0] + fv[1]) + (fv[2] + fv[3]);
            const float vark = totk * (1.f / kD) + eps_k, varv = totv * (1.f / kD) + eps_v;
            const float rstdk = __builtin_amdgcn_rsqf(vark), rstdv = __builtin_amdgcn_rsqf(varv);
            float sigma = varv * rstdv;
            asm volatile("" : "+v"(sigma));                                  // one fp32 value for both halves (see retr_attn.hip, p2_store)
            const _Float16 sh = (_Float16)sigma;
            const _Float16 sl = (_Float16)(sigma - (float)sh);
            const _Float16 one = (_Float16)1.0f;
            const uint32_t w0 = (uint32_t)__builtin_bit_cast(uint16_t, one) | ((uint32_t)__builtin_bit_cast(uint16_t, sh) << 16);
            const uint32_t w1 = (uint32_t)__builtin_bit_cast(uint16_t, sl);
            const u32x4 row16 = {w0, w1, __float_as_uint(rstdk), __float_as_uint(rstdv)};
            asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" : : "v"(row16), "v"(aoff), "s"(asrd) : "memory");
        }
        {   // this wave's pieces of tile it+2 have landed: everything but the batches it+3 .. it+kA-1 and the aux stores issued since
            wait_vm<nb * (kA - 3) + nst * (kA - 2)>();
            if (!A.map_f16) {
                u32x4 w0_ = *reinterpret_cast<SVPS_LDS const u32x4*>((uintptr_t)(cv_lane + off_c));
                u32x4 w1_ = *reinterpret_cast<SVPS_LDS const u32x4*>((uintptr_t)(cv_lane + off_c + 1024));
                convert_piece(off_c, w0_, 0);
                convert_piece(off_c, w1_, 1);
            }
        }
        dma_batch(off_d, yoff_d, px0_d, yrow_d, it + kA < nt);
        off_l = ring_next(off_l);
        off_c = ring_next(off_c);
        off_d = ring_next(off_d);
        yoff_l = yring_next(yoff_l);
        yoff_d = yring_next(yoff_d);
        px0_d += W;
        ++yrow_d;
    }
    wait_vm<0>();
}

__global__ __launch_bounds__(512) void retr_stats2_kernel(const Args A) {
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    switch (w) {                 // every role runs the same sequence of workgroup barriers
        case 0: role<0, 0>(A); break;
        case 1: role<0, 1>(A); break;
        case 2: role<0, 2>(A); break;
        case 3: role<0, 3>(A); break;
        case 4: role<1, 0>(A); break;
        case 5: role<1, 1>(A); break;
        case 6: role<1, 2>(A); break;
        default: role<1, 3>(A); break;
    }
}

}  // namespace s2
}  // namespace svps

namespace {
struct PlanS2 {
    int strips, cps, tpc;
    int chunks() const { return strips * cps; }
};
// chunks never leave a strip; rounds of one workgroup per CU; every workgroup pays a prologue of about six tile times
PlanS2 plan_s2(int T, int H, int W) {
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
                                         void* const* aux, int T, int H, int W, int D, int flags, void* stream_) {
    if (!feat || !ty || !tx || !rk || !rbk || !lnk_eps || !rv || !rbv || !lnv_eps || !aux) return SVPS_ERR_BAD_ARG;
    if (n_stages < 1 || n_stages > 2) return SVPS_ERR_BAD_SHAPE;
    if (D != svps::kD || T <= 0 || H <= 0 || W <= 0) return SVPS_ERR_BAD_SHAPE;
    if ((size_t)H * W > svps::kMaxFramePixels) return SVPS_ERR_BAD_SHAPE;
    for (int s = 0; s < n_stages; ++s)
        if (!ty[s] || !tx[s] || !rk[s] || !rbk[s] || !rv[s] || !rbv[s] || !aux[s]) return SVPS_ERR_BAD_ARG;
    if (n_stages == 1)           // a level with a single stage: the per-stage kernel (retr_stats.hip)
        return svps_retr_stats_fwd(feat, ty[0], tx[0], rk[0], rbk[0], lnk_eps[0], rv[0], rbv[0], lnv_eps[0], aux[0], T, H, W, D, flags, stream_);
    svps::s2::Args a;
    a.feat = static_cast<const __bf16*>(feat);
    for (int s = 0; s < 2; ++s) {
        a.st[s].ty = ty[s]; a.st[s].tx = tx[s];
        a.st[s].rk = static_cast<const _Float16*>(rk[s]); a.st[s].rv = static_cast<const _Float16*>(rv[s]);
        a.st[s].rbk = rbk[s]; a.st[s].rbv = rbv[s];
        a.st[s].aux = static_cast<__bf16*>(aux[s]);
        a.st[s].eps_k = lnk_eps[s]; a.st[s].eps_v = lnv_eps[s];
    }
    const PlanS2 p = plan_s2(T, H, W);
    a.HW = H * W; a.H = H; a.W = W; a.tiles_per_chunk = p.tpc; a.chunks_per_strip = p.cps;
    a.map_f16 = (flags & SVPS_FLAG_MAP_F16) ? 1 : 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    static SvpsLdsAttr attr;
    if (hipError_t ae = attr.ensure(reinterpret_cast<const void*>(svps::s2::retr_stats2_kernel), svps::s2::Lds::total); ae != hipSuccess) return (int)ae;
    svps_prof_mark(SVPS_KERNEL_RETR_STATS, 0, stream);
    hipLaunchKernelGGL(svps::s2::retr_stats2_kernel, dim3(p.chunks(), T), dim3(512), svps::s2::Lds::total, stream, a);
    svps_prof_mark(SVPS_KERNEL_RETR_STATS, 1, stream);
    return (int)hipGetLastError();
}
