// K1'' - ONE pass over the fused map per stage: LayerNorm statistics + logits + softmax over slots + P.f in one kernel (gfx950).
//
// MaskDynamicConv.forward (mmdet/models/detectors/dynamic_mask_head.py:423-461). The algebra is that of retr_stats.hip (K3') and
// retr_attn.hip (K1'): both pixel-side LayerNorms are "one scalar per pixel times an affine map",
//     rstd_k(p) = 1 / sqrt(|R_k f_p + Ty[y] + Tx[x] + r_k|^2 / 256 + eps)      rstd_v(p) = 1 / sqrt(|R_v f_p + r_v|^2 / 256 + eps)
//     S[l, p]   = rstd_k(p) (Q''_l . f_p + Cy[y, l] + Cx[x, l]) + c3_l          P = softmax over the SLOT axis (:446)
//     A_l = sum_p P rstd_v f_p,   s1_l = sum_p P rstd_v,   s0_l = sum_p P       (:456; W~_v, norm1, ReLU follow on the slot side)
// K3' + K1' read the map twice per stage and hand the two statistics through HBM (16 B per pixel). Here a workgroup keeps ALL of
// it on the CU: the statistics of a tile cross its four waves through LDS, nothing per pixel is written, the map is read once.
//
// Shape: 4 waves = one per SIMD, 512 registers each (__launch_bounds__(256); VGPR + AGPR are one file on gfx950). Wave sb owns
//   * slot block sb: Q'' hi / lo (fp16, 128 registers), the accumulator block A[32 sb .. +32, 0:256] + the aux block (s1, s0): 144
//   * row blocks (sb, 7 - sb) of BOTH upper-triangular factors R_k, R_v: 2 x 18 fragments = 144 registers
// = 416 registers of matrix state per lane. The position tables of the strip (Tx + r_k: 32 x 256, Cx: 32 x 128 fp32) live in LDS.
// hipcc's own allocation spills at this pressure (it keeps every MFMA operand in arch VGPRs), so every MFMA is an asm statement
// with explicit register classes: A accumulators, R_v and half of Q'' lo in AGPRs (248), everything the vector ALU touches in VGPRs.
//
// Tile = 32 consecutive pixels of one image row; a workgroup walks DOWN a 32-pixel-wide column strip and never leaves it (the
// planner cuts chunks inside strips), so every x-dependent term is loaded once per workgroup. 5-deep LDS ring of 16-KiB tiles
// (LDS-DMA, swizzled on the source side; each wave converts the four 1-KiB pieces it requested bf16 -> fp16 in place).
//
// Pipeline, ONE workgroup barrier per tile. Iteration `it` (after barrier B(it)):
//     logits(it)     32 MFMA  Q'' hi, lo x row fragments of tile it          || finish(it-1): P rstd_v = e * fac -> fp16 -> LDS
//     stats1(it+1)   2 NK0    R_k, R_v row block sb x row fragments it+1     || head(it): * rstd_k + c3, block max, exp2, block sum
//     PVa(it-1)       9 MFMA  A += P f, pixels 0..15 (+ aux block)           || sum of squares of stats1, init of stats2
//     stats2(it+1)   2 NK1    row block 7 - sb                               || bf16 -> fp16 of the own pieces of tile it+2
//     PVb(it-1)       9 MFMA  pixels 16..31                                  || sum of squares of stats2 -> LDS
// = 86 MFMA 32x32x16 per tile and SIMD. Exchanges through LDS: sums of squares of tile it+1 (read in iteration it+1), block
// (max, sum) of tile it (read by finish(it) in iteration it+1). The exponentials of a tile stay in 16 registers across the barrier.
// The vector work is cut into steps that are placed, by hand, in the shadow of the MFMAs (sched_barrier fences between slots).
#include <type_traits>

#include "common.h"
#include "../../include/slotvps_hip.h"

namespace svps {
namespace fz {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(2))) __fp16 fp16x2_t;
typedef __fp16 fp16x4_gcc __attribute__((__vector_size__(4 * sizeof(__fp16))));

constexpr int kFN = 5;                 // feature ring depth: tiles it-1 (P.f), it (logits), it+1 (statistics), it+2 (landing / conversion), it+3 (in flight)
constexpr int kYSlot = 2048;           // per ring slot: Ty row (1 KiB) | Cy piece (1 KiB: the tile's Cy row, 512 B, + the next row)
constexpr int kTxRow = 1040;           // bytes per pixel row of the Tx table (256 floats + 16: conflict-free 16-B reads across pixel rows)
constexpr int kCxRow = 528;            // bytes per pixel row of the Cx table (128 floats + 16)
constexpr int kPartRow = 260;          // floats per slot row of a partial (retr_attn.hip)
constexpr int kQlA = 10;               // k-steps of Q'' lo that live in AGPRs (with R_v and the A accumulators: all 256)

struct Lds {
    static constexpr int fring = 0;                          // kFN x 16 KiB (tile bases are multiples of 512 B: fragment address XORs)
    static constexpr int yring = fring + kFN * kTileBytes;   // kFN x 2 KiB
    static constexpr int txt = yring + kFN * kYSlot;         // [32 px][256] fp32  Tx[x] + r_k
    static constexpr int cxt = txt + 32 * kTxRow;            // [32 px][128] fp32  Cx[x]
    static constexpr int pbuf = cxt + 32 * kCxRow;           // 4 waves x 2 KiB: P rstd_v of the wave's slot block, [32 px][32 slots] fp16
    static constexpr int auxb = pbuf + 4 * 2048;             // [2][32 px] 16-byte aux rows {1, hi sigma_v, lo sigma_v, 0, 0, 0, 0, 0} fp16
    static constexpr int x1 = auxb + 2 * 512;                // [2][2 proj][32 px][4 waves] fp32 sums of squares
    static constexpr int x2 = x1 + 2 * 2 * 32 * 4 * 4;       // [2][32 px][4 waves] float2 (block max, block sum)
    static constexpr int rbv = x2 + 2 * 32 * 4 * 8;          // [256] fp32  r_v
    static constexpr int c3 = rbv + 1024;                    // [128] fp32
    static constexpr int total = c3 + 512;
};
static_assert(Lds::total <= 160 * 1024 && Lds::pbuf % 16 == 0, "LDS layout");

#define FZ_FENCE() __builtin_amdgcn_sched_barrier(0)

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}

// MFMA slots with the vector steps [i * NSTEP / NS, (i + 1) * NSTEP / NS) behind slot i; no slots: the steps alone.
template <int NS, int NSTEP, class M, class St>
__device__ __forceinline__ void phase(M&& mfma, St&& step) {
    if constexpr (NS == 0) {
        sfor<0, NSTEP>(step);
    } else {
        sfor<0, NS>([&](auto I) {
            constexpr int i = decltype(I)::value;
            mfma(I);
            FZ_FENCE();
            sfor<i * NSTEP / NS, (i + 1) * NSTEP / NS>(step);
            FZ_FENCE();
        });
    }
}

// v_mfma_f32_32x32x16_f16 with explicit register classes. hipcc pads no hazard inside (or around) an asm statement:
//   * "s_nop 1" in front: an operand a vector instruction has just written (accumulator initialisation, a register copy the
//     compiler may have placed) needs two wait states before the matrix instruction reads it
//   * a reader of an accumulator other than the next MFMA of its chain waits with fz_mfma_settle() (19 states)
__device__ __forceinline__ void mfma_vvv(f32x16& acc, const f16x8& a, const f16x8& b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_vav(f32x16& acc, const f16x8& a, const f16x8& b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
}
__device__ __forceinline__ void mfma_avv(f32x16& acc, const f16x8& a, const f16x8& b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void settle(f32x16& x) { asm volatile("s_nop 15\n\ts_nop 2" : "+v"(x)); }
__device__ __forceinline__ void settle2(f32x16& x, f32x16& y) { asm volatile("s_nop 15\n\ts_nop 2" : "+v"(x), "+v"(y)); }

__device__ __forceinline__ u32x4 make_srd(const void* base, uint32_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    u32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
    d[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
    d[2] = __builtin_amdgcn_readfirstlane(bytes);
    d[3] = 0x00020000u;
    return d;
}
// asm LDS-DMA (the builtin form makes hipcc drain the ring before every LDS read: slot_attn.hip); `nt`: the map is read once
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

__device__ __forceinline__ float half_swap_max(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_swap_sum(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ f32x4 lds4(uint32_t a) { return *reinterpret_cast<SVPS_LDS const f32x4*>((uintptr_t)a); }
__device__ __forceinline__ f16x8 lds8h(uint32_t a) { return *reinterpret_cast<SVPS_LDS const f16x8*>((uintptr_t)a); }

struct Args {
    const _Float16* qh;      // [T, 128, 256] hi(Q'')
    const _Float16* ql;      // [T, 128, 256] lo(Q'')
    const float* cy;         // [T, H, 128]
    const float* cx;         // [T, W, 128]
    const float* c3g;        // [T, 128]  log2(e) q . beta_k; -1e30 in the padded rows
    const __bf16* feat;      // [T, HW, 256]
    const float* ty;         // [H, 256]  R_k[:, :128] ytab[y]
    const float* tx;         // [W, 256]  R_k[:, 128:] xtab[x]
    const _Float16* rk;      // [256, 256] fp16 upper triangular
    const _Float16* rv;
    const float* rbk;        // [256]
    const float* rbv;
    float eps_k, eps_v;
    float* partial;          // [T, C, L, 260]
    int L, HW, H, W, tiles_per_chunk, chunks_per_strip;
};

template <int SB>
__device__ __forceinline__ void role(const Args& A) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int LP = 128;
    constexpr int rb0 = SB, rb1 = 7 - SB;
    constexpr int NK0 = 2 * (8 - SB), NK1 = 2 * (SB + 1);          // k-steps of the two row blocks (18 fragments per factor)
    constexpr int sb = SB;
    const int lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const int C = gridDim.x;
    int t = blockIdx.y, c = blockIdx.x;
    if ((gridDim.y & 7) == 0) {
        // XCD-aware frame placement (speed only): all chunks of a frame on ONE XCD, so its L2 holds that frame's slot operands
        const int b = blockIdx.y * C + blockIdx.x;
        const int n = b >> 3;
        t = (b & 7) + 8 * (n / C);
        c = n % C;
    }
    const int H = A.H, W = A.W, HW = A.HW;
    const float eps_k = A.eps_k, eps_v = A.eps_v;
    const int strip = c / A.chunks_per_strip;
    const int y0 = (c - strip * A.chunks_per_strip) * A.tiles_per_chunk;
    int nt = H - y0;
    nt = nt < A.tiles_per_chunk ? nt : A.tiles_per_chunk;           // >= 1 by construction of the grid
    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>((SVPS_LDS const void*)smem);
    const int x0 = kTilePx * strip;
    const bool live = x0 + r < W;                                   // pixels past the right edge of the map: P = 0

    // ---- tables of the strip and small vectors -> LDS --------------------------------------------------------------------
    {
        const int tid = threadIdx.x;
        const float rb = A.rbk[tid];
        float* txl = reinterpret_cast<float*>(smem + Lds::txt);
#pragma unroll 4
        for (int px = 0; px < 32; ++px) {
            int xx = x0 + px;
            xx = xx < W ? xx : W - 1;
            txl[px * (kTxRow / 4) + tid] = A.tx[(size_t)xx * kD + tid] + rb;
        }
        float* cxl = reinterpret_cast<float*>(smem + Lds::cxt);
        const int sl = tid & 127, pp = tid >> 7;
#pragma unroll 4
        for (int p2 = 0; p2 < 16; ++p2) {
            const int px = 2 * p2 + pp;
            int xx = x0 + px;
            xx = xx < W ? xx : W - 1;
            cxl[px * (kCxRow / 4) + sl] = A.cx[((size_t)t * W + xx) * LP + sl];
        }
        reinterpret_cast<float*>(smem + Lds::rbv)[tid] = A.rbv[tid];
        if (tid < 128) reinterpret_cast<float*>(smem + Lds::c3)[tid] = A.c3g[(size_t)t * LP + tid];
    }
    FZ_FENCE();
    // ---- resident matrix operands -------------------------------------------------------------------------------------
    // Loaded and pinned to their register class in small groups (a fence after each): with all 272 registers of loads in flight
    // at once hipcc runs out of arch VGPRs HERE and then spills the operands for the whole kernel. AGPR: R_v and the first half of
    // Q'' lo; VGPR: the rest. The pin also makes hipcc wait for every load here: its wait-count pass does not see the asm waits of
    // the main loop and would otherwise drain the LDS-DMA ring inside it.
    f16x8 qfh[16], qfl[16], wk0[NK0], wk1[NK1], wv0[NK0], wv1[NK1];
    {
        const _Float16* qrow_h = A.qh + ((size_t)t * LP + 32 * sb + r) * kD + 8 * h;
        const _Float16* qrow_l = A.ql + ((size_t)t * LP + 32 * sb + r) * kD + 8 * h;
        const size_t w0off = (size_t)(32 * rb0 + r) * kD + 32 * rb0 + 8 * h, w1off = (size_t)(32 * rb1 + r) * kD + 32 * rb1 + 8 * h;
        sfor<0, NK0>([&](auto I) {
            constexpr int i = decltype(I)::value;
            wv0[i] = *reinterpret_cast<const f16x8*>(A.rv + w0off + 16 * i);
            asm volatile("" : "+a"(wv0[i]));
            if constexpr ((i & 3) == 3) FZ_FENCE();
        });
        FZ_FENCE();
        sfor<0, NK1>([&](auto I) {
            constexpr int i = decltype(I)::value;
            wv1[i] = *reinterpret_cast<const f16x8*>(A.rv + w1off + 16 * i);
            asm volatile("" : "+a"(wv1[i]));
            if constexpr ((i & 3) == 3) FZ_FENCE();
        });
        FZ_FENCE();
        sfor<0, kQlA>([&](auto I) {
            constexpr int i = decltype(I)::value;
            qfl[i] = *reinterpret_cast<const f16x8*>(qrow_l + 16 * i);
            asm volatile("" : "+a"(qfl[i]));
            if constexpr ((i & 3) == 3) FZ_FENCE();
        });
        FZ_FENCE();
        sfor<kQlA, 16>([&](auto I) {
            constexpr int i = decltype(I)::value;
            qfl[i] = *reinterpret_cast<const f16x8*>(qrow_l + 16 * i);
            asm volatile("" : "+v"(qfl[i]));
            if constexpr ((i & 3) == 3) FZ_FENCE();
        });
        sfor<0, 16>([&](auto I) {
            constexpr int i = decltype(I)::value;
            qfh[i] = *reinterpret_cast<const f16x8*>(qrow_h + 16 * i);
            asm volatile("" : "+v"(qfh[i]));
            if constexpr ((i & 3) == 3) FZ_FENCE();
        });
        sfor<0, NK0>([&](auto I) {
            constexpr int i = decltype(I)::value;
            wk0[i] = *reinterpret_cast<const f16x8*>(A.rk + w0off + 16 * i);
            asm volatile("" : "+v"(wk0[i]));
            if constexpr ((i & 3) == 3) FZ_FENCE();
        });
        FZ_FENCE();
        sfor<0, NK1>([&](auto I) {
            constexpr int i = decltype(I)::value;
            wk1[i] = *reinterpret_cast<const f16x8*>(A.rk + w1off + 16 * i);
            asm volatile("" : "+v"(wk1[i]));
            if constexpr ((i & 3) == 3) FZ_FENCE();
        });
        FZ_FENCE();
    }
    wait_vm<0>();

    f32x16 o[8], oa;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        oa[i] = 0.f;
#pragma unroll
        for (int db = 0; db < 8; ++db) o[db][i] = 0.f;
    }
#pragma unroll
    for (int db = 0; db < 8; ++db) asm volatile("" : "+a"(o[db]));
    asm volatile("" : "+a"(oa));

    // ---- LDS-DMA: wave sb stages rows 8 sb .. 8 sb + 7 of every tile (4 pieces); wave 0 the Ty row, wave 2 the Cy piece ------------
    const u32x4 frs = make_srd(A.feat + (size_t)t * HW * kD, (uint32_t)HW * kRowBytes);
    const u32x4 tys = make_srd(A.ty, (uint32_t)H * 1024u);
    const u32x4 cys = make_srd(A.cy + (size_t)t * H * LP, (uint32_t)(H * LP) * 4u);
    constexpr int nb = 4 + ((SB == 0 || SB == 2) ? 1 : 0);           // DMA instructions of one batch of this wave
    auto piece_voff = [&](int i, int px0) {
        const int row = 8 * sb + 2 * i + h;
        const int src = px0 + row < HW ? row : HW - 1 - px0;          // last image row of a ragged strip: clamp (those pixels are not live)
        return src * kRowBytes + (((lane & 31) ^ swz(row)) * 16);
    };
    auto dma_piece = [&](int b, int i) {                              // piece i (0 .. 3: features, 4: table row) of batch b
        if (b >= nt) return;
        const int slot = b % kFN;
        const int px0 = (y0 + b) * W + x0;
        if (i < 4) {
            const uint32_t st = __builtin_amdgcn_readfirstlane(lds0 + Lds::fring + slot * kTileBytes + sb * 4096 + i * 1024);
            dma16_nt(frs, st, piece_voff(i, px0), __builtin_amdgcn_readfirstlane(px0 * kRowBytes));
        } else if (SB == 0) {
            dma16(tys, __builtin_amdgcn_readfirstlane(lds0 + Lds::yring + slot * kYSlot), lane * 16,
                  __builtin_amdgcn_readfirstlane((y0 + b) * 1024));
        } else if (SB == 2) {
            dma16(cys, __builtin_amdgcn_readfirstlane(lds0 + Lds::yring + slot * kYSlot + 1024), lane * 16,
                  __builtin_amdgcn_readfirstlane((y0 + b) * LP * 4));
        }
    };
    // this wave's four pieces of tile b: bf16 -> fp16 in place (exact for |f| in [6.1e-5, 65504]; retr_attn.hip)
    u32x4 cvA, cvB;
    auto convert_load = [&](int b, int i, u32x4& w_) {
        const uint32_t st = lds0 + Lds::fring + (b % kFN) * kTileBytes + sb * 4096 + lane * 16;
        w_ = *reinterpret_cast<SVPS_LDS const u32x4*>((uintptr_t)(st + i * 1024));
    };
    auto convert_store = [&](int b, int i, u32x4& w_) {
        const uint32_t st = lds0 + Lds::fring + (b % kFN) * kTileBytes + sb * 4096 + lane * 16;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const fp16x2_t pk = __builtin_amdgcn_cvt_pkrtz(__uint_as_float(w_[k] << 16), __uint_as_float(w_[k] & 0xffff0000u));
            w_[k] = __builtin_bit_cast(uint32_t, pk);
        }
        *reinterpret_cast<SVPS_LDS u32x4*>((uintptr_t)(st + i * 1024)) = w_;
    };

    // ---- fragment addressing (retr_attn.hip) ----------------------------------------------------------------------------
    const uint32_t lane_row = lds0 + Lds::fring + r * kRowBytes + ((h ^ swz(r)) << 4);
    auto frag = [&](uint32_t tb, int ks) { return lds8h((tb ^ ((ks & 7) << 5)) + 256 * (ks >> 3)); };
    const int g2 = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    const int cl = 2 * (g2 & 1) + (pp >> 1), sub = 8 * (pp & 1), rowl = 8 * (g2 >> 1) + qq;
    const uint32_t lane_v0 = rowl * kRowBytes + (((cl ^ (2 * (g2 >> 1))) + 4 * qq) << 4) + sub;
    const uint32_t lane_v1 = (rowl + 4) * kRowBytes + (((cl ^ (2 * (g2 >> 1) + 1)) + 4 * qq) << 4) + sub;
    const uint32_t lane_p0 = lds0 + Lds::pbuf + sb * 2048 + sub + rowl * 64 + ((cl ^ (qq >> 1)) << 4);
    const uint32_t lane_p1 = lds0 + Lds::pbuf + sb * 2048 + sub + (rowl + 4) * 64 + ((cl ^ (qq >> 1) ^ 2) << 4);
    const uint32_t lane_a = rowl * 16 + ((cl == 0 && sub != 0) ? 8 : 0);
    auto tr = [](uint32_t a) {
        return __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16(reinterpret_cast<SVPS_LDS fp16x4_gcc*>((uintptr_t)a)));
    };
    auto cat = [](f16x4 a, f16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); };

    // ---- state that crosses iterations ------------------------------------------------------------------------------------
    f32x16 e;                                   // logits of tile it, then its exponentials (relative to the block maximum)
    float mloc_p = 0.f, tau_p = 0.f;            // block maximum / rstd_v of the tile whose exponentials `e` holds
    f16x8 fb[4];                                // row-fragment ring: fragment F of an iteration lives in fb[F % 4], three ahead
#pragma unroll
    for (int i = 0; i < 16; ++i) e[i] = 0.f;

    const int slot0 = 32 * sb + 4 * h;          // accumulator register 4 g + j <-> slot row slot0 + 8 g + j
    const int key = (r >> 1) & 3;

    // One iteration. ST: statistics of tile it+1 exist; LG: tile it exists (logits + softmax head); FN: tile it-1 exists
    // (softmax finish + P.f).
    auto body = [&](int it, auto st_tag, auto lg_tag, auto fn_tag) {
        constexpr bool ST = decltype(st_tag)::value, LG = decltype(lg_tag)::value, FN = decltype(fn_tag)::value;
        const uint32_t slot_s = (uint32_t)((it + 1 + kFN) % kFN), slot_l = (uint32_t)((it + kFN) % kFN), slot_f = (uint32_t)((it - 1 + kFN) % kFN);
        const uint32_t tb_l = lane_row + slot_l * kTileBytes, tb_s = lane_row + slot_s * kTileBytes;
        const uint32_t par_l = it & 1, par_s = (it + 1) & 1, par_f = (it - 1) & 1;
        const bool more_dma = it + 3 < nt;
        const bool fin_ok = it >= 1;                                 // tile it-1 exists (it <= nt by the loop)
        // fragment stream of the iteration: [logits 0 .. 15][statistics row block rb0: k-steps 2 rb0 ..][row block rb1: 2 rb1 ..]
        constexpr int FL = LG ? 16 : 0, F0 = ST ? NK0 : 0, F1 = ST ? NK1 : 0, NFR = FL + F0 + F1;
        auto load_frag = [&](auto Fi) {
            constexpr int f = decltype(Fi)::value;
            if constexpr (f < NFR) {
                if constexpr (f < FL) fb[f % 4] = frag(tb_l, f);
                else if constexpr (f < FL + F0) fb[f % 4] = frag(tb_s, 2 * rb0 + (f - FL));
                else fb[f % 4] = frag(tb_s, 2 * rb1 + (f - FL - F0));
            }
        };

        // ================= phase 1: logits(it) || finish(it-1) ==========================================================
        f32x16 s;
        f32x4 stA, stB;                          // finish: (max, sum) of the four slot blocks for this lane's pixel
        float fac = 0.f, mall = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
        f32x4 xk, xv;                            // sums of squares of tile it (four waves), for rstd_k / rstd_v
        float rk_c = 0.f, tau_c = 0.f;
        f32x16 ak, av;                           // statistics accumulators (row block rb0, then rb1)
        if constexpr (FN) {
            const uint32_t a = lds0 + Lds::x2 + par_f * 1024 + r * 32;
            stA = lds4(a);
            stB = lds4(a + 16);
        }
        if constexpr (LG) {
            // (the first three fragments were requested before the barrier)
            const uint32_t cya = lds0 + Lds::yring + slot_l * kYSlot + 1024 + slot0 * 4;
            const uint32_t cxa = lds0 + Lds::cxt + r * kCxRow + slot0 * 4;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 a = lds4(cya + 32 * g), b = lds4(cxa + 32 * g);
#pragma unroll
                for (int j = 0; j < 4; ++j) s[4 * g + j] = a[j] + b[j];
            }
            const uint32_t xa = lds0 + Lds::x1 + par_l * 1024 + r * 16;
            xk = lds4(xa);
            xv = lds4(xa + 512);
        }
        char* prow = smem + Lds::pbuf + sb * 2048 + r * 64 + 8 * h;
        auto p1_step = [&](auto K) {
            constexpr int k = decltype(K)::value;
            if constexpr (k == 0) { dma_piece(it + 3, 0); }
            if constexpr (FN) {
                if constexpr (k == 1) mall = fmaxf(fmaxf(stA[0], stA[2]), fmaxf(stB[0], stB[2]));
                if constexpr (k == 2) { d0 = __builtin_amdgcn_exp2f(stA[0] - mall); d1 = __builtin_amdgcn_exp2f(stA[2] - mall); }
                if constexpr (k == 3) { d2 = __builtin_amdgcn_exp2f(stB[0] - mall); d3 = __builtin_amdgcn_exp2f(stB[2] - mall); }
                if constexpr (k == 4) {
                    const float den = (stA[1] * d0 + stA[3] * d1) + (stB[1] * d2 + stB[3] * d3);
                    fac = __builtin_amdgcn_exp2f(mloc_p - mall) * __builtin_amdgcn_rcpf(den) * tau_p;
                    if (!live) fac = 0.f;
                }
                if constexpr (k >= 5 && k < 9) {                    // four slots of P(it-1) rstd_v = e * fac -> fp16
                    constexpr int g = k - 5;
                    f16x4 ph;
#pragma unroll
                    for (int j = 0; j < 4; ++j) ph[j] = (_Float16)(e[4 * g + j] * fac);
                    *reinterpret_cast<f16x4*>(prow + ((g ^ key) * 16)) = ph;
                }
            }
            if constexpr (k == 9) { dma_piece(it + 3, 1); }
            if constexpr (LG) {
                if constexpr (k == 10) {                            // rstd_k, rstd_v of tile it
                    const float totk = (xk[0] + xk[1]) + (xk[2] + xk[3]), totv = (xv[0] + xv[1]) + (xv[2] + xv[3]);
                    const float vark = totk * (1.f / kD) + eps_k, varv = totv * (1.f / kD) + eps_v;
                    rk_c = __builtin_amdgcn_rsqf(vark) * kLog2e;
                    tau_c = __builtin_amdgcn_rsqf(varv);
                    if constexpr (SB == 1) {                        // the aux row of the pixel (one writer per workgroup)
                        const float sigma = varv * tau_c;
                        const _Float16 sh = (_Float16)sigma;
                        const _Float16 sl = (_Float16)(sigma - (float)sh);
                        const _Float16 one = (_Float16)1.0f;
                        const uint32_t w0 = (uint32_t)__builtin_bit_cast(uint16_t, one) | ((uint32_t)__builtin_bit_cast(uint16_t, sh) << 16);
                        const uint32_t w1 = (uint32_t)__builtin_bit_cast(uint16_t, sl);
                        if (h == 0) *reinterpret_cast<u32x4*>(smem + Lds::auxb + par_l * 512 + r * 16) = u32x4{w0, w1, 0u, 0u};
                    }
                }
            }
            if constexpr (k == 11) { dma_piece(it + 3, 2); }
        };
        constexpr int NP1 = 12;
        phase<LG ? 32 : 0, NP1>(
            [&](auto I) {
                constexpr int i = decltype(I)::value, f = i >> 1;
                if constexpr ((i & 1) == 0) {
                    mfma_vvv(s, qfh[f], fb[f % 4]);
                } else {
                    if constexpr (f < kQlA) mfma_vav(s, qfl[f], fb[f % 4]);
                    else mfma_vvv(s, qfl[f], fb[f % 4]);
                    load_frag(std::integral_constant<int, f + 3>{});
                }
            },
            p1_step);

        // ================= phase 2: stats1(it+1) || head(it) =============================================================
        if constexpr (ST) {
            const uint32_t ys = lds0 + Lds::yring + slot_s * kYSlot + (32 * rb0 + 4 * h) * 4;
            const uint32_t txa = lds0 + Lds::txt + r * kTxRow + (32 * rb0 + 4 * h) * 4;
            const uint32_t rva = lds0 + Lds::rbv + (32 * rb0 + 4 * h) * 4;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 a = lds4(ys + 32 * g), b = lds4(txa + 32 * g), cvv = lds4(rva + 32 * g);
#pragma unroll
                for (int j = 0; j < 4; ++j) { ak[4 * g + j] = a[j] + b[j]; av[4 * g + j] = cvv[j]; }
            }
            if constexpr (!LG) { sfor<0, 3>(load_frag); }
        }
        float mloc = kNegBig, sloc = 0.f;
        f32x4 c3v;
        auto head_step = [&](auto K) {
            constexpr int k = decltype(K)::value;
            if constexpr (LG) {
                if constexpr (k == 0) settle(s);
                if constexpr (k >= 1 && k < 5) {                    // log2(e) S = (log2(e) rstd_k) (Q''.f + Cy + Cx) + c3'
                    constexpr int g = k - 1;
                    c3v = lds4(lds0 + Lds::c3 + (slot0 + 8 * g) * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) s[4 * g + j] = fmaf(rk_c, s[4 * g + j], c3v[j]);
                }
                if constexpr (k == 5) {
                    mloc = fmaxf(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])), fmaxf(fmaxf(s[4], s[5]), fmaxf(s[6], s[7])));
                }
                if constexpr (k == 6) {
                    const float m2 = fmaxf(fmaxf(fmaxf(s[8], s[9]), fmaxf(s[10], s[11])), fmaxf(fmaxf(s[12], s[13]), fmaxf(s[14], s[15])));
                    mloc = half_swap_max(fmaxf(mloc, m2));
                }
                if constexpr (k >= 7 && k < 15) {                   // two exponentials per step
                    constexpr int i0 = 2 * (k - 7);
                    s[i0] = __builtin_amdgcn_exp2f(s[i0] - mloc);
                    s[i0 + 1] = __builtin_amdgcn_exp2f(s[i0 + 1] - mloc);
                }
                if constexpr (k == 15) sloc = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
                if constexpr (k == 16) {
                    sloc += ((s[8] + s[9]) + (s[10] + s[11])) + ((s[12] + s[13]) + (s[14] + s[15]));
                    sloc = half_swap_sum(sloc);
                    if (h == 0) *reinterpret_cast<float2*>(smem + Lds::x2 + par_l * 1024 + r * 32 + sb * 8) = make_float2(mloc, sloc);
                    e = s;
                    mloc_p = mloc;
                    tau_p = tau_c;
                }
            }
            if constexpr (k == 17) { dma_piece(it + 3, 3); }
        };
        constexpr int NHD = 18;
        phase<ST ? 2 * NK0 : 0, NHD>(
            [&](auto I) {
                constexpr int i = decltype(I)::value, j = i >> 1, f = FL + j;
                if constexpr ((i & 1) == 0) {
                    mfma_vav(av, wv0[j], fb[f % 4]);
                } else {
                    mfma_vvv(ak, wk0[j], fb[f % 4]);
                    load_frag(std::integral_constant<int, f + 3>{});
                }
            },
            head_step);

        // ================= phase 3: PVa(it-1) || sums of squares of stats1, init of stats2 =================================
        float sqk = 0.f, sqv = 0.f, sqk2 = 0.f, sqv2 = 0.f;
        f16x8 ah, af, vf[4];
        uint32_t p0 = 0, p1 = 0, v0 = 0, v1 = 0, aa = 0;
        if constexpr (FN) {
            p0 = lane_p0, p1 = lane_p1;
            const uint32_t vt = lds0 + Lds::fring + slot_f * kTileBytes;
            v0 = vt + lane_v0, v1 = vt + lane_v1;
            aa = lds0 + Lds::auxb + par_f * 512 + lane_a;
        }
        auto vfrag = [&](int ks, int db) {
            const uint32_t o_ = 8192 * ks + 256 * (db >> 2);
            return cat(tr((v0 ^ ((db & 3) << 6)) + o_), tr((v1 ^ ((db & 3) << 6)) + o_));
        };
        auto pv_slot = [&](auto KS, auto I) {                       // slot i of 9: channel blocks 0 .. 7, then the aux block
            constexpr int ks = decltype(KS)::value, i = decltype(I)::value;
            if constexpr (i < 8) {
                if (fin_ok) mfma_avv(o[i], ah, vf[i % 4]);
                if constexpr (i + 3 < 8) vf[(i + 3) % 4] = vfrag(ks, i + 3);
                if constexpr (i + 3 == 8) af = cat(tr(aa + 16 * 16 * ks), tr(aa + 16 * 16 * ks + 4 * 16));
            } else {
                if (fin_ok) mfma_avv(oa, ah, af);
            }
        };
        auto pv_begin = [&](int ks) {
            ah = cat(tr(p0 + 1024 * ks), tr(p1 + 1024 * ks));
#pragma unroll
            for (int u = 0; u < 3; ++u) vf[u] = vfrag(ks, u);
        };
        auto sq_step = [&](auto K) {                                // sums of squares of (ak, av) over this lane's 16 rows: 8 steps
            constexpr int k = decltype(K)::value;
            if constexpr (k < 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) sqk = fmaf(ak[4 * k + j], ak[4 * k + j], sqk);
            } else if constexpr (k < 8) {
#pragma unroll
                for (int j = 0; j < 4; ++j) sqv = fmaf(av[4 * (k - 4) + j], av[4 * (k - 4) + j], sqv);
            }
        };
        f32x4 ta, tbv;
        auto p3_step = [&](auto K) {
            constexpr int k = decltype(K)::value;
            if constexpr (ST) {
                if constexpr (k == 0) settle2(ak, av);
                if constexpr (k >= 1 && k < 9) sq_step(std::integral_constant<int, k - 1>{});
                if constexpr (k == 9) { sqk2 = sqk; sqv2 = sqv; sqk = 0.f; sqv = 0.f; }
                if constexpr (k >= 10 && k < 14) {                  // accumulators of row block rb1
                    constexpr int g = k - 10;
                    ta = lds4(lds0 + Lds::yring + slot_s * kYSlot + (32 * rb1 + 8 * g + 4 * h) * 4);
                    tbv = lds4(lds0 + Lds::txt + r * kTxRow + (32 * rb1 + 8 * g + 4 * h) * 4);
                    const f32x4 cvv = lds4(lds0 + Lds::rbv + (32 * rb1 + 8 * g + 4 * h) * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { ak[4 * g + j] = ta[j] + tbv[j]; av[4 * g + j] = cvv[j]; }
                }
            }
        };
        if constexpr (FN) pv_begin(0);
        phase<FN ? 9 : 0, 14>([&](auto I) { pv_slot(std::integral_constant<int, 0>{}, I); }, p3_step);

        // ================= phase 4: stats2(it+1) || conversion of the own pieces of tile it+2 ==================================
        const bool cv = it + 2 < nt;
        auto p4_step = [&](auto K) {
            constexpr int k = decltype(K)::value;
            if constexpr (k == 0) { if constexpr (nb == 5) dma_piece(it + 3, 4); }
            if constexpr (k == 1) {
                if (cv) {
                    // landed: batch it+2 - everything but the batch requested in this iteration
                    if (more_dma) wait_vm<nb>(); else wait_vm<0>();
                    convert_load(it + 2, 0, cvA);
                    convert_load(it + 2, 1, cvB);
                }
            }
            if constexpr (k == 2) { if (cv) { convert_store(it + 2, 0, cvA); convert_load(it + 2, 2, cvA); } }
            if constexpr (k == 3) { if (cv) { convert_store(it + 2, 1, cvB); convert_load(it + 2, 3, cvB); } }
            if constexpr (k == 4) { if (cv) convert_store(it + 2, 2, cvA); }
            if constexpr (k == 5) { if (cv) convert_store(it + 2, 3, cvB); }
        };
        phase<ST ? 2 * NK1 : 0, 6>(
            [&](auto I) {
                constexpr int i = decltype(I)::value, j = i >> 1, f = FL + F0 + j;
                if constexpr ((i & 1) == 0) {
                    mfma_vav(av, wv1[j], fb[f % 4]);
                } else {
                    mfma_vvv(ak, wk1[j], fb[f % 4]);
                    load_frag(std::integral_constant<int, f + 3>{});
                }
            },
            p4_step);

        // ================= phase 5: PVb(it-1) || sums of squares of stats2 -> LDS ============================================
        auto p5_step = [&](auto K) {
            constexpr int k = decltype(K)::value;
            if constexpr (ST) {
                if constexpr (k == 0) settle2(ak, av);
                if constexpr (k >= 1 && k < 9) sq_step(std::integral_constant<int, k - 1>{});
                if constexpr (k == 9) {
                    const float totk = half_swap_sum(sqk + sqk2), totv = half_swap_sum(sqv + sqv2);
                    if (h == 0) {
                        float* x1l = reinterpret_cast<float*>(smem + Lds::x1 + par_s * 1024);
                        x1l[r * 4 + sb] = totk;
                        x1l[128 + r * 4 + sb] = totv;
                    }
                }
            }
        };
        if constexpr (FN) pv_begin(1);
        phase<FN ? 9 : 0, 10>([&](auto I) { pv_slot(std::integral_constant<int, 1>{}, I); }, p5_step);
    };

    using T_ = std::true_type;
    using F_ = std::false_type;
    // ---- prologue: batches 0, 1 requested; tile 0 landed, converted, published -----------------------------------------------
#pragma unroll
    for (int i = 0; i < 5; ++i) { if (i < nb) dma_piece(0, i); }
#pragma unroll
    for (int i = 0; i < 5; ++i) { if (i < nb) dma_piece(1, i); }
    if (nt > 1) wait_vm<nb>(); else wait_vm<0>();
#pragma unroll
    for (int i = 0; i < 4; ++i) { convert_load(0, i, cvA); convert_store(0, i, cvA); }
    wg_barrier();                                                    // B(-1): tile 0 is fp16, the tables are in LDS
    // ONE body for every iteration, it = -1 .. nt (statistics of tile it+1, logits of tile it, finish + P.f of tile it-1): the
    // iterations at either end run their MFMAs on tiles that do not exist (stale or uninitialised LDS). Nothing of that reaches a
    // result: statistics / block sums of a missing tile are only read by the head / finish of that same missing tile, and the
    // P.f MFMAs - the only writers of the result - are skipped when tile it-1 does not exist. Six specialised instantiations of the
    // body (first version) made hipcc spill every resident operand; two extra iterations per chunk are the cheaper price.
    auto prefetch_logits = [&](int it) {                             // its tile was published one barrier ago
        const uint32_t tb = lane_row + (uint32_t)((it + kFN) % kFN) * kTileBytes;
#pragma unroll
        for (int f = 0; f < 3; ++f) fb[f] = frag(tb, f);
    };
    for (int it = -1; it <= nt; ++it) {
        prefetch_logits(it);
        wg_barrier();                                                // B(it)
        body(it, T_{}, T_{}, T_{});
    }

    // ---- partial sums of this chunk -> HBM ------------------------------------------------------------------------------------
    asm volatile("s_nop 15\n\ts_nop 2" ::: "memory");
    float* dst = A.partial + ((size_t)t * C + c) * A.L * kPartRow;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int slot = 32 * sb + acc_row(i, h);
        if (slot < A.L) {
#pragma unroll
            for (int db = 0; db < 8; ++db) dst[(size_t)slot * kPartRow + 32 * db + r] = o[db][i];
            if (r < 4) dst[(size_t)slot * kPartRow + 256 + r] = oa[i];
        }
    }
}

__global__ __launch_bounds__(256) void retr_fused_kernel(const Args A) {
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef FZ_ONLY_ROLE
    role<FZ_ONLY_ROLE>(A);
#else
    switch (w) {                 // every role runs the same sequence of workgroup barriers
        case 0: role<0>(A); break;
        case 1: role<1>(A); break;
        case 2: role<2>(A); break;
        default: role<3>(A); break;
    }
#endif
}

}  // namespace fz

// Sum of the C partials of every (frame, slot) row in chunk order (bitwise reproducible, no float atomics); out row (272 floats) =
// { A[0:256], s1, s0, 0 x 14 }: the operand of the slot-side product (the kernel of the same name in retr_attn.hip, for this file's grid)
namespace fz {
__global__ __launch_bounds__(256) void fused_finish_kernel(const float* __restrict__ partial, float* __restrict__ out, int L, int C) {
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
    float* o = out + ((size_t)t * L + l) * 272;
    o[d] = colsum(d);
    if (d < 16) {
        float v = 0.f;
        if (d == 0) v = colsum(256);                         // s1 = sum_p P rstd_v
        else if (d == 1) v = colsum(257) + colsum(258);      // s0 = sum_p P  (sigma_v carried as hi + lo)
        o[256 + d] = v;
    }
}
}  // namespace fz

}  // namespace svps

namespace {
struct FusedPlan {
    int strips, cps, tpc;        // column strips, chunks per strip, tiles (image rows) per chunk
    int chunks() const { return strips * cps; }
};
// Chunks never leave a strip. All workgroups do nearly the same work, so the launch runs in rounds of one workgroup per CU: take the
// split whose last round is fullest, charging every workgroup its prologue (operands, tables, pipeline fill: about six tile times).
FusedPlan plan_fused(int T, int H, int W, int cps_req) {
    const int strips = (W + svps::kTilePx - 1) / svps::kTilePx;
    int cps = cps_req;
    if (cps <= 0) {
        const int cus = svps_num_cus();
        double best = -1.0;
        cps = 1;
        for (int c = 1; c <= H; ++c) {
            const int tpc = (H + c - 1) / c;
            if (c > 1 && tpc < 8) break;
            const int cc = (H + tpc - 1) / tpc;
            const long wg = (long)T * strips * cc;
            const long rounds = (wg + cus - 1) / cus;
            const double eff = (double)T * strips * H / ((double)rounds * cus * (tpc + 6));
            if (eff > best + 1e-9) { best = eff; cps = cc; }
        }
    }
    if (cps > H) cps = H;
    const int tpc = (H + cps - 1) / cps;
    cps = (H + tpc - 1) / tpc;
    return {strips, cps, tpc};
}
}  // namespace

extern "C" size_t svps_retr_fused_workspace_bytes(int T, int L, int H, int W, int chunks_per_strip) {
    if (T <= 0 || L <= 0 || H <= 0 || W <= 0) return 0;
    const FusedPlan p = plan_fused(T, H, W, chunks_per_strip);
    return (size_t)T * p.chunks() * L * svps::fz::kPartRow * sizeof(float);
}

extern "C" int svps_retr_fused_fwd(const void* qh, const void* ql, const float* cy, const float* cx, const float* c3,
                                   const void* feat, const float* ty, const float* tx, const void* rk, const float* rbk, float lnk_eps,
                                   const void* rv, const float* rbv, float lnv_eps, void* workspace, size_t workspace_bytes,
                                   float* out_ext, int T, int L, int H, int W, int D, int chunks_per_strip, void* stream_) {
    if (!qh || !ql || !cy || !cx || !c3 || !feat || !ty || !tx || !rk || !rbk || !rv || !rbv || !workspace || !out_ext) return SVPS_ERR_BAD_ARG;
    if (D != svps::kD || T <= 0 || L <= 0 || L > 128 || H <= 0 || W <= 0) return SVPS_ERR_BAD_SHAPE;
    if ((size_t)H * W > svps::kMaxFramePixels) return SVPS_ERR_BAD_SHAPE;
    const FusedPlan p = plan_fused(T, H, W, chunks_per_strip);
    const size_t need = (size_t)T * p.chunks() * L * svps::fz::kPartRow * sizeof(float);
    if (workspace_bytes < need) return SVPS_ERR_WORKSPACE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    svps::fz::Args a;
    a.qh = static_cast<const _Float16*>(qh);
    a.ql = static_cast<const _Float16*>(ql);
    a.cy = cy; a.cx = cx; a.c3g = c3;
    a.feat = static_cast<const __bf16*>(feat);
    a.ty = ty; a.tx = tx;
    a.rk = static_cast<const _Float16*>(rk);
    a.rv = static_cast<const _Float16*>(rv);
    a.rbk = rbk; a.rbv = rbv;
    a.eps_k = lnk_eps; a.eps_v = lnv_eps;
    a.partial = static_cast<float*>(workspace);
    a.L = L; a.HW = H * W; a.H = H; a.W = W; a.tiles_per_chunk = p.tpc; a.chunks_per_strip = p.cps;
    static SvpsLdsAttr attr;
    if (hipError_t ae = attr.ensure(reinterpret_cast<const void*>(svps::fz::retr_fused_kernel), svps::fz::Lds::total); ae != hipSuccess) return (int)ae;
    svps_prof_mark(SVPS_KERNEL_RETR_FUSED, 0, stream);
    hipLaunchKernelGGL(svps::fz::retr_fused_kernel, dim3(p.chunks(), T), dim3(256), svps::fz::Lds::total, stream, a);
    hipError_t e = hipGetLastError();
    svps_prof_mark(SVPS_KERNEL_RETR_FUSED, 1, stream);
    if (e != hipSuccess) return (int)e;
    svps_prof_mark(SVPS_KERNEL_RETR_FINISH, 0, stream);
    hipLaunchKernelGGL(svps::fz::fused_finish_kernel, dim3(L, T), dim3(256), 0, stream, static_cast<const float*>(workspace), out_ext, L, p.chunks());
    svps_prof_mark(SVPS_KERNEL_RETR_FINISH, 1, stream);
    return (int)hipGetLastError();
}
