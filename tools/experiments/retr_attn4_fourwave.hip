// K1''' - the statistics-fused retriever as FOUR waves, one per SIMD, 512 registers each (gfx950).
//
// MaskDynamicConv.forward (mmdet/models/detectors/dynamic_mask_head.py:423-461); same function, inputs and outputs as
// retr_attn_kernel (retr_attn.hip, K1'):
//     S[l, p] = rstd_k(p) (Q''_l . f_p + Cy[y, l] + Cx[x, l]) + c3_l       P = softmax over the SLOT axis (:446)
//     A_l = sum_p P rstd_v f_p,   s1_l = sum_p P rstd_v,   s0_l = sum_p P   (:456; W~_v, norm1, ReLU follow on the slot side)
// with rstd_k, rstd_v from the 16-byte aux rows of the statistics kernel (retr_stats.hip / retr_stats4.hip).
//
// Why a second form. K1' runs 8 waves (producer + consumer per SIMD) and sits at a third of either roofline: its two waves per SIMD
// serialise on the age-arbitrated matrix pipe inside a one-barrier-per-tile workgroup. Here ONE wave per SIMD
// owns a slot block end to end - Q'' hi / lo (128 registers, lo in AGPRs), the logits, the softmax, P through a wave-private LDS
// transpose, and the accumulator block A[32 sb .. +32, 0:256] + the aux block (144 AGPRs): 50 MFMA 32x32x16 per tile in ONE
// in-order stream, the vector work cut into steps that are placed by hand in the shadow of the MFMAs (sched_barrier fences
// between slots). Every MFMA is an asm statement with explicit register classes (hipcc otherwise reads the logit accumulator
// back through v_accvgpr_read and copies operands between the files).
//
// Tile = 32 consecutive pixels of one image row; a workgroup walks DOWN a 32-pixel-wide column strip and never leaves it (the
// planner cuts chunks inside strips): the Cx terms of the strip sit in a 16.5-KiB LDS table, loaded once per workgroup.
// 7-deep ring of 16-KiB tiles (LDS-DMA, swizzled on the source side; each wave converts the four 1-KiB pieces it requested
// bf16 -> fp16 in place), aux tile + Cy row staged with every tile.
//
// Pipeline, ONE workgroup barrier per tile. Iteration `it` (after barrier B(it)):
//     logits(it)   32 MFMA  Q'' hi, lo x row fragments of tile it      || finish(it-1): P rstd_v = e * fac -> fp16 -> LDS;
//                                                                         bf16 -> fp16 of the own pieces of tile it+2; DMA of tile it+5
//     P.f(it-1)    18 MFMA  A += P f (+ aux block)                       || head(it): * rstd_k + c3, block max, exp2, block sum
// The block (max, sum) pairs of tile it cross the four waves through LDS (read by finish(it) in iteration it+1); the
// exponentials of a tile stay in 16 registers across the barrier. The first row fragments of tile it+1 are requested BEFORE
// barrier B(it+1) (its pieces were converted one iteration earlier), so the matrix pipe starts right behind the barrier.
#include <type_traits>

#include "common.h"
#include "../../include/slotvps_hip.h"

namespace svps {
namespace r4 {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(2))) __fp16 fp16x2_t;
typedef __fp16 fp16x4_gcc __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#ifndef R4_QL_AGPR
#define R4_QL_AGPR 0        // 1: Q'' lo in AGPRs (measured: an MFMA whose A operand sits in an AGPR next to a VGPR accumulator is slower)
#endif
#ifndef R4_KEEP
#define R4_KEEP 1
#endif
#ifndef R4_FB
#define R4_FB 6             // row-fragment ring: R4_FB - 1 fragments ahead
#endif

constexpr int kFN = 7;                 // ring depth: tiles it-1 (P.f), it (logits), it+1 (fp16), it+2 (converting), it+3 .. it+5 (in flight)
constexpr int kA = kFN - 2;            // batch it + kA is requested in iteration it
constexpr int kAuxRow = 16;            // bytes per pixel of the aux tensor
constexpr int kCxRow = 528;            // bytes per pixel row of the Cx table (128 floats + 16: conflict-free 16-byte reads across pixel rows)
constexpr int kPartRow = 260;          // floats per slot row of a partial: 256 channels of A + 4 aux columns
constexpr int kExtRow = 272;           // floats per slot row of the finished result
constexpr int kFB = R4_FB;
constexpr int kVR = 8;                 // P.f operand ring (hardware-transposed fragments): kVR - 1 ahead of the MFMA that uses them

struct Lds {
    static constexpr int fring = 0;                          // kFN x 16 KiB (tile bases are multiples of 512 B: fragment address XORs)
    static constexpr int aring = fring + kFN * kTileBytes;   // kFN x 1 KiB: aux tile (32 rows of 16 B; the upper half of the DMA piece repeats them)
    static constexpr int yring = aring + kFN * 1024;         // kFN x 1 KiB: the tile's Cy row (512 B) + the next row
    static constexpr int cxt = yring + kFN * 1024;           // [32 px][128] fp32 Cx[x]
    static constexpr int pbuf = cxt + 32 * kCxRow;           // 4 waves x 2 KiB: P rstd_v of the wave's slot block, [32 px][32 slots] fp16
    static constexpr int x2 = pbuf + 4 * 2048;               // [2][32 px][4 waves] float2 (block max, block sum)
    static constexpr int c3 = x2 + 2 * 32 * 4 * 8;           // [128] fp32
    static constexpr int total = c3 + 512;
};
static_assert(Lds::total <= 160 * 1024 && Lds::pbuf % 16 == 0, "LDS layout");

#define R4_FENCE() __builtin_amdgcn_sched_barrier(0)

// timing-only ablations of a diagnostic build (make abl4; results wrong): 1 no LDS-DMA after the prologue  2 no conversion
// 4 no softmax head  8 no softmax finish  16 no P.f MFMAs  32 no logit MFMAs  64 no P.f operand reads
#ifndef R4_ABL
#define R4_ABL 0
#endif

#ifdef R4_STAMP
// diagnostic build only (make stamp4, tools/retr4_stamps.py): s_memtime stamps of the four waves of ONE workgroup, iterations
// 8 .. 15, kept in LDS behind the kernel's own data (a global store per stamp would count in vmcnt) and copied out at the end;
// plus (s_memtime, s_memrealtime) around the loop of every workgroup's wave 0 for the in-kernel clock
__device__ unsigned long long r4_stamps[4][8][8];       // [wave][iteration - 8][point]
__device__ unsigned long long r4_clock[4096][4];        // [workgroup][memtime0, realtime0, memtime1, realtime1]
#define R4_STAMP_AT(pt)                                                                                          \
    do {                                                                                                         \
        R4_FENCE();                                                                                              \
        if (stamp_wg && it >= 8 && it < 16) {                                                                    \
            unsigned long long t_;                                                                               \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                           \
            if (lane == 0) reinterpret_cast<unsigned long long*>(smem + Lds::total)[(sb * 8 + (it - 8)) * 8 + pt] = t_; \
        }                                                                                                        \
        R4_FENCE();                                                                                              \
    } while (0)
#else
#define R4_STAMP_AT(pt) do {} while (0)
#endif

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}

// NS MFMA slots with the vector steps [i * NSTEP / NS, (i + 1) * NSTEP / NS) behind slot i
template <int NS, int NSTEP, class M, class St>
__device__ __forceinline__ void phase(M&& mfma, St&& step) {
    sfor<0, NS>([&](auto I) {
        constexpr int i = decltype(I)::value;
        mfma(I);
        R4_FENCE();
        sfor<i * NSTEP / NS, (i + 1) * NSTEP / NS>(step);
        R4_FENCE();
    });
}

// v_mfma_f32_32x32x16_f16 with explicit register classes. hipcc pads no hazard inside (or around) an asm statement:
//   * "s_nop 1" in front of the first MFMA of the logit chain: its accumulator has just been written by vector adds and needs two
//     wait states before the matrix instruction reads it (every other operand of the loop comes out of an LDS read or an MFMA:
//     check the ISA for compiler-placed v_mov copies in front of an asm MFMA after an edit)
//   * a reader of an accumulator other than the next MFMA of its chain first passes settle() (19 wait states)
__device__ __forceinline__ void mfma_vvv_head(f32x16& acc, const f16x8& a, const f16x8& b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_vvv(f32x16& acc, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_vav(f32x16& acc, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
}
__device__ __forceinline__ void mfma_avv(f32x16& acc, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void settle(f32x16& x) { asm volatile("s_nop 15\n\ts_nop 2" : "+v"(x)); }

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
    const __bf16* aux;       // [T, HW, 8] 16-byte rows {1, hi sigma_v, lo sigma_v, 0 (fp16), rstd_k, rstd_v (fp32)}
    float* partial;          // [T, C, L, 260]
    int L, HW, H, W, tiles_per_chunk, chunks_per_strip;
};

template <int SB>
__device__ __forceinline__ void role(const Args& A) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int LP = 128;
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
    const int strip = c / A.chunks_per_strip;
    const int y0 = (c - strip * A.chunks_per_strip) * A.tiles_per_chunk;
    int nt = H - y0;
    nt = nt < A.tiles_per_chunk ? nt : A.tiles_per_chunk;           // >= 1 by construction of the grid
    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>((SVPS_LDS const void*)smem);
#ifdef R4_STAMP
    const bool stamp_wg = blockIdx.x == 3 && blockIdx.y == 2;
#endif
    const int x0 = kTilePx * strip;
    const bool live = x0 + r < W;                                   // pixels past the right edge of the map: P = 0

    // ---- the strip's Cx table and c3 -> LDS ----------------------------------------------------------------------------------
    {
        const int tid = threadIdx.x;
        float* cxl = reinterpret_cast<float*>(smem + Lds::cxt);
        const int sl = tid & 127, pp = tid >> 7;
#pragma unroll 4
        for (int p2 = 0; p2 < 16; ++p2) {
            const int px = 2 * p2 + pp;
            int xx = x0 + px;
            xx = xx < W ? xx : W - 1;
            cxl[px * (kCxRow / 4) + sl] = A.cx[((size_t)t * W + xx) * LP + sl];
        }
        if (tid < 128) reinterpret_cast<float*>(smem + Lds::c3)[tid] = A.c3g[(size_t)t * LP + tid];
    }
    R4_FENCE();
    // ---- resident operands: Q'' hi in VGPRs, Q'' lo in AGPRs. Loaded and pinned in small groups, every load waited for HERE: hipcc's
    // wait-count pass does not see the asm waits of the main loop and would otherwise drain the LDS-DMA ring inside it.
    f16x8 qfh[16], qfl[16];
    {
        const _Float16* qrow_h = A.qh + ((size_t)t * LP + 32 * sb + r) * kD + 8 * h;
        const _Float16* qrow_l = A.ql + ((size_t)t * LP + 32 * sb + r) * kD + 8 * h;
        sfor<0, 16>([&](auto I) {
            constexpr int i = decltype(I)::value;
            qfl[i] = *reinterpret_cast<const f16x8*>(qrow_l + 16 * i);
#if R4_QL_AGPR
            asm volatile("" : "+a"(qfl[i]));
#else
            asm volatile("" : "+v"(qfl[i]));
#endif
            if constexpr ((i & 3) == 3) R4_FENCE();
        });
        sfor<0, 16>([&](auto I) {
            constexpr int i = decltype(I)::value;
            qfh[i] = *reinterpret_cast<const f16x8*>(qrow_h + 16 * i);
            asm volatile("" : "+v"(qfh[i]));
            if constexpr ((i & 3) == 3) R4_FENCE();
        });
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

    // ---- LDS-DMA: wave sb stages rows 8 sb .. 8 sb + 7 of every tile (4 pieces); wave 0 the aux tile, wave 2 the Cy row ---------------
    // A lone wave issues one instruction per ~4 cycles whatever its kind, so the loop below is BRANCH-FREE: every iteration requests a
    // batch, converts a tile and runs all 50 MFMAs. A batch past the end of the chunk is requested through a descriptor of ZERO
    // records (the hardware range check returns zeros, no memory traffic) into a ring slot nobody reads; rows past the end of the
    // frame (last image row of a ragged strip) are cut by the same check - those pixels are not live. Constant vmcnt waits follow.
    constexpr int nb = 4 + ((SB == 0 || SB == 2) ? 1 : 0);           // DMA instructions of one batch of this wave
    const uint64_t fbase = reinterpret_cast<uint64_t>(A.feat + (size_t)t * HW * kD);
    const uint64_t xbase = SB == 0 ? reinterpret_cast<uint64_t>(A.aux + (size_t)t * HW * 8) : reinterpret_cast<uint64_t>(A.cy + (size_t)t * H * LP);
    const uint32_t fs0 = __builtin_amdgcn_readfirstlane((uint32_t)fbase), fs1 = __builtin_amdgcn_readfirstlane((uint32_t)(fbase >> 32) & 0xffffu);
    const uint32_t xs0 = __builtin_amdgcn_readfirstlane((uint32_t)xbase), xs1 = __builtin_amdgcn_readfirstlane((uint32_t)(xbase >> 32) & 0xffffu);
    const uint32_t frec = (uint32_t)HW * kRowBytes, xrec = SB == 0 ? (uint32_t)HW * kAuxRow : (uint32_t)(H * LP) * 4u;
    int voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * sb + 2 * i + h;
        voff[i] = row * kRowBytes + (((lane & 31) ^ swz(row)) * 16);
    }
    const int xvoff = SB == 0 ? (lane & 31) * kAuxRow : lane * 16;
    // batch state (wave-uniform): ring offset of the batch's slot, first pixel of its tile, whether the tile exists
    auto dma_piece = [&](int i, uint32_t off_d, int px0_d, int yrow_d, bool ok) {
        if (i < 4) {
            const u32x4 srd = {fs0, fs1, ok ? frec : 0u, 0x00020000u};
            dma16_nt(srd, lds0 + Lds::fring + off_d + sb * 4096 + i * 1024, voff[i], px0_d * kRowBytes);
        } else if (SB == 0) {
            const u32x4 srd = {xs0, xs1, ok ? xrec : 0u, 0x00020000u};
            dma16(srd, lds0 + Lds::aring + (off_d >> 4), xvoff, px0_d * kAuxRow);
        } else if (SB == 2) {
            const u32x4 srd = {xs0, xs1, ok ? xrec : 0u, 0x00020000u};
            dma16(srd, lds0 + Lds::yring + (off_d >> 4), xvoff, yrow_d * (LP * 4));
        }
    };
    auto ring_next = [](uint32_t off) { return off + kTileBytes == (uint32_t)kFN * kTileBytes ? 0u : off + kTileBytes; };
    // this wave's four pieces of the tile at ring offset `off`: bf16 -> fp16 in place (exact for |f| in [6.1e-5, 65504]; retr_attn.hip)
    const uint32_t cv_lane = lds0 + Lds::fring + sb * 4096 + lane * 16;
    u32x4 cvw[4];
    auto convert_load = [&](uint32_t off, int i) { cvw[i] = *reinterpret_cast<SVPS_LDS const u32x4*>((uintptr_t)(cv_lane + off + i * 1024)); };
    auto convert_store = [&](uint32_t off, int i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const fp16x2_t pk = __builtin_amdgcn_cvt_pkrtz(__uint_as_float(cvw[i][k] << 16), __uint_as_float(cvw[i][k] & 0xffff0000u));
            cvw[i][k] = __builtin_bit_cast(uint32_t, pk);
        }
        *reinterpret_cast<SVPS_LDS u32x4*>((uintptr_t)(cv_lane + off + i * 1024)) = cvw[i];
    };

    // ---- fragment addressing (retr_attn.hip) ----------------------------------------------------------------------------
    const uint32_t lane_row = lds0 + Lds::fring + r * kRowBytes + ((h ^ swz(r)) << 4);
    auto frag = [&](uint32_t tb, int ks) { return lds8h((tb ^ ((ks & 7) << 5)) + 256 * (ks >> 3)); };
    const int g2 = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    const int cl = 2 * (g2 & 1) + (pp >> 1), sub = 8 * (pp & 1), rowl = 8 * (g2 >> 1) + qq;
    const uint32_t lane_v0 = lds0 + Lds::fring + rowl * kRowBytes + (((cl ^ (2 * (g2 >> 1))) + 4 * qq) << 4) + sub;
    const uint32_t lane_v1 = lds0 + Lds::fring + (rowl + 4) * kRowBytes + (((cl ^ (2 * (g2 >> 1) + 1)) + 4 * qq) << 4) + sub;
    // value fragments: chunk (4 db + cl) ^ swz(row): the (db & 3) part is one of four XOR patterns on bits 6 .. 7 of the lane address
    // (tile bases are multiples of 16 KiB), (db >> 2) and the k-step are instruction offsets
    uint32_t lv0[4], lv1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { lv0[k] = lane_v0 ^ (k << 6); lv1[k] = lane_v1 ^ (k << 6); }
    const uint32_t lane_p0 = lds0 + Lds::pbuf + sb * 2048 + sub + rowl * 64 + ((cl ^ (qq >> 1)) << 4);
    const uint32_t lane_p1 = lds0 + Lds::pbuf + sb * 2048 + sub + (rowl + 4) * 64 + ((cl ^ (qq >> 1) ^ 2) << 4);
    // aux block: the row's FP16 words 0 .. 3 are columns 0 .. 3 of the ninth channel block; the lanes that feed the other 28 columns
    // read words 4 .. 7 (the two fp32 statistics as bit patterns) or repeat words 0 .. 3: those columns only reach accumulator
    // columns nobody stores
    const uint32_t lane_a = lds0 + Lds::aring + rowl * kAuxRow + ((cl == 0 && sub != 0) ? 8 : 0);
    auto tr = [](uint32_t a) {
        return __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16(reinterpret_cast<SVPS_LDS fp16x4_gcc*>((uintptr_t)a)));
    };
    auto cat = [](f16x4 a, f16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); };

    // ---- state that crosses iterations ------------------------------------------------------------------------------------
    f32x16 e;                                   // exponentials of the previous tile (relative to its block maximum)
    float mloc_p = 0.f, tau_p = 0.f;            // block maximum / rstd_v of the tile whose exponentials `e` holds
    f16x8 fb[kFB];                              // row-fragment ring: fragment f lives in fb[f % kFB], kFB - 1 ahead
#pragma unroll
    for (int i = 0; i < 16; ++i) e[i] = 0.f;
    const int slot0 = 32 * sb + 4 * h;          // accumulator register 4 g + j <-> slot row slot0 + 8 g + j
    const int key = (r >> 1) & 3;
    const uint32_t prow = lds0 + Lds::pbuf + sb * 2048 + r * 64 + 8 * h;
    const uint32_t cx_lane = lds0 + Lds::cxt + r * kCxRow + slot0 * 4, cy_lane = lds0 + Lds::yring + slot0 * 4;
    const uint32_t rt_lane = lds0 + Lds::aring + r * kAuxRow + 8, x2_lane = lds0 + Lds::x2 + r * 32;

    // ONE body for every iteration, it = 0 .. nt (logits + softmax head of tile it, finish + P.f of tile it-1). The last iteration
    // runs its logits on a tile that does not exist (stale LDS): its exponentials are never finished. Iteration 0 runs the P.f MFMAs
    // of a tile that does not exist either: with P = 0 (e = 0 and a finite factor) against a zero-filled tile.
    // off_l / off_f: ring offsets of tiles it / it-1; off_c: tile it+2 (converted here); off_d, px0_d, yrow_d: batch it+kA.
    f32x16 s;                                   // logits of tile it: starts from Cy + Cx (set before the barrier)
    f32x2 rt = {0.f, 0.f};                      // (rstd_k, rstd_v) of this lane's pixel in tile it
    auto body = [&](int it, uint32_t off_l, uint32_t off_f, uint32_t off_c, uint32_t off_d, int px0_d, int yrow_d) {
        const uint32_t tb_l = lane_row + off_l;
        const uint32_t par_l = (it & 1) * 1024, par_f = 1024 - par_l;
        const bool dma_ok = it + kA < nt;
        auto load_frag = [&](auto Fi) {
            constexpr int f = decltype(Fi)::value;
            if constexpr (f < 16) fb[f % kFB] = frag(tb_l, f);
        };

        R4_STAMP_AT(0);
        // ================= phase 1: logits(it) || finish(it-1), conversion of tile it+2, DMA of tile it+kA ======================
        float fac = 0.f, mall = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
        // finish: (max, sum) of the four slot blocks for this lane's pixel (written by the other waves before the barrier)
        const f32x4 stA = lds4(x2_lane + par_f), stB = lds4(x2_lane + par_f + 16);
        f16x8 ah0, ah1, af, vf[kVR];
        uint32_t vx0[4], vx1[4];
        const uint32_t aa = lane_a + (off_f >> 4);
        auto vfrag = [&](int v) {                                   // v = 8 ks + db
            const int ks = v >> 3, db = v & 7;
            const uint32_t o_ = 8192 * ks + 256 * (db >> 2);
            return cat(tr(vx0[db & 3] + o_), tr(vx1[db & 3] + o_));
        };
        auto p1_step = [&](auto K) {
            constexpr int k = decltype(K)::value;
            if constexpr (!((R4_ABL & 8) && k < 8)) {
                if constexpr (k == 0) mall = fmaxf(fmaxf(stA[0], stA[2]), fmaxf(stB[0], stB[2]));
                if constexpr (k == 1) { d0 = __builtin_amdgcn_exp2f(stA[0] - mall); d1 = __builtin_amdgcn_exp2f(stA[2] - mall); }
                if constexpr (k == 2) { d2 = __builtin_amdgcn_exp2f(stB[0] - mall); d3 = __builtin_amdgcn_exp2f(stB[2] - mall); }
                if constexpr (k == 3) {
                    const float den = (stA[1] * d0 + stA[3] * d1) + (stB[1] * d2 + stB[3] * d3);
                    fac = __builtin_amdgcn_exp2f(mloc_p - mall) * __builtin_amdgcn_rcpf(den) * tau_p;
                    if (!live) fac = 0.f;
                }
                if constexpr (k >= 4 && k < 8) {                    // four slots of P(it-1) rstd_v = e * fac -> fp16
                    constexpr int g = k - 4;
                    f16x4 ph;
#pragma unroll
                    for (int j = 0; j < 4; ++j) ph[j] = (_Float16)(e[4 * g + j] * fac);
                    *reinterpret_cast<SVPS_LDS u32x2*>((uintptr_t)(prow + ((g ^ key) * 16))) = __builtin_bit_cast(u32x2, ph);
                }
            }
            if constexpr (k == 8 && !(R4_ABL & 2)) {
                // landed: this wave's pieces of batch it+2 = everything but the batches it+3 .. it+kA-1 (this iteration's come later)
                wait_vm<nb * (kA - 3)>();
#pragma unroll
                for (int i = 0; i < 4; ++i) convert_load(off_c, i);
            }
            if constexpr (k >= 10 && k < 14 && !(R4_ABL & 2)) convert_store(off_c, k - 10);
            if constexpr (!(R4_ABL & 1)) {
                if constexpr (k == 9) dma_piece(0, off_d, px0_d, yrow_d, dma_ok);
                if constexpr (k >= 14 && k < 17) dma_piece(k - 13, off_d, px0_d, yrow_d, dma_ok);
                if constexpr (k == 17 && nb == 5) dma_piece(4, off_d, px0_d, yrow_d, dma_ok);
            }
            if constexpr (k == 18 && !(R4_ABL & 64)) {              // first operands of P.f(it-1)
#pragma unroll
                for (int q = 0; q < 4; ++q) { vx0[q] = lv0[q] + off_f; vx1[q] = lv1[q] + off_f; }
                ah0 = cat(tr(lane_p0), tr(lane_p1));
#pragma unroll
                for (int u = 0; u < kVR - 1; ++u) vf[u] = vfrag(u);
            }
        };
        phase<32, 19>(
            [&](auto I) {
                constexpr int i = decltype(I)::value, f = i >> 1;
                if constexpr (!(R4_ABL & 32)) {
                    if constexpr (i == 0) mfma_vvv_head(s, qfh[0], fb[0]);       // s has just been written by vector adds
                    else if constexpr ((i & 1) == 0) mfma_vvv(s, qfh[f], fb[f % kFB]);
#if R4_QL_AGPR
                    else mfma_vav(s, qfl[f], fb[f % kFB]);
#else
                    else mfma_vvv(s, qfl[f], fb[f % kFB]);
#endif
                }
                if constexpr (i & 1) {
#if R4_KEEP
                    // keep the fragment of the PREVIOUS pair alive past this pair: hipcc otherwise loads the next fragment into the
                    // registers the MFMA issued just above is still reading, and the load waits for that MFMA
                    if constexpr (f >= 1) asm volatile("" ::"v"(fb[(f - 1) % kFB]));
                    load_frag(std::integral_constant<int, f + kFB - 2>{});
#else
                    load_frag(std::integral_constant<int, f + kFB - 1>{});
#endif
                }
#ifdef R4_STAMP
                if constexpr (i == 15) R4_STAMP_AT(1);
#endif
            },
            p1_step);
        R4_STAMP_AT(2);

        // ================= phase 2: P.f(it-1) || head(it) ===================================================================
        const float rk_c = rt[0] * kLog2e, tau_c = rt[1] * kPScale;          // common.h: the probabilities carry 2^7
        float mloc = kNegBig, sloc = 0.f;
        f32x4 c3v[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) c3v[g] = lds4(lds0 + Lds::c3 + (slot0 + 8 * g) * 4);
        auto head_step = [&](auto K) {
            constexpr int k = decltype(K)::value;
            if constexpr (R4_ABL & 4) return;
            if constexpr (k == 0) settle(s);
            if constexpr (k >= 1 && k < 5) {                        // log2(e) S = (log2(e) rstd_k) (Q''.f + Cy + Cx) + c3'
                constexpr int g = k - 1;
#pragma unroll
                for (int j = 0; j < 4; ++j) s[4 * g + j] = fmaf(rk_c, s[4 * g + j], c3v[g][j]);
            }
            if constexpr (k == 5) mloc = fmaxf(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])), fmaxf(fmaxf(s[4], s[5]), fmaxf(s[6], s[7])));
            if constexpr (k == 6) {
                const float m2 = fmaxf(fmaxf(fmaxf(s[8], s[9]), fmaxf(s[10], s[11])), fmaxf(fmaxf(s[12], s[13]), fmaxf(s[14], s[15])));
                mloc = half_swap_max(fmaxf(mloc, m2));
            }
            if constexpr (k >= 7 && k < 15) {                       // two exponentials per step
                constexpr int i0 = 2 * (k - 7);
                s[i0] = __builtin_amdgcn_exp2f(s[i0] - mloc);
                s[i0 + 1] = __builtin_amdgcn_exp2f(s[i0 + 1] - mloc);
            }
            if constexpr (k == 15) sloc = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
            if constexpr (k == 16) {
                sloc += ((s[8] + s[9]) + (s[10] + s[11])) + ((s[12] + s[13]) + (s[14] + s[15]));
                sloc = half_swap_sum(sloc);
                if (h == 0) *reinterpret_cast<SVPS_LDS f32x2*>((uintptr_t)(x2_lane + par_l + sb * 8)) = f32x2{mloc, sloc};
                e = s;
                mloc_p = mloc;
                tau_p = tau_c;
            }
        };
        phase<18, 17>(
            [&](auto I) {                                           // slots 0 .. 8: pixels 0 .. 15 (channel blocks 0 .. 7, aux block); 9 .. 17: pixels 16 .. 31
                constexpr int i = decltype(I)::value, ks = i / 9, j = i % 9, v = 8 * ks + j;
                if constexpr (j < 8) {
                    if constexpr (!(R4_ABL & 16)) mfma_avv(o[j], ks ? ah1 : ah0, vf[v % kVR]);
                    if constexpr (!(R4_ABL & 64)) {
                        if constexpr (v + kVR - 1 < 16) vf[(v + kVR - 1) % kVR] = vfrag(v + kVR - 1);
                        if constexpr (j == 1) af = cat(tr(aa + 16 * kAuxRow * ks), tr(aa + 16 * kAuxRow * ks + 4 * kAuxRow));
                        if constexpr (i == 2) ah1 = cat(tr(lane_p0 + 1024), tr(lane_p1 + 1024));
                    }
                } else {
                    if constexpr (!(R4_ABL & 16)) mfma_avv(oa, ks ? ah1 : ah0, af);
                }
#ifdef R4_STAMP
                if constexpr (i == 8) R4_STAMP_AT(3);
#endif
            },
            head_step);
        R4_STAMP_AT(4);
    };

    // ---- prologue: the slot of "tile -1" and its aux rows zero-filled, the block statistics set to (0, 1); batches 0 .. kA-1
    // requested; tiles 0 and 1 landed, converted, published -------------------------------------------------------------------
    {
        const int tid = threadIdx.x;
        u32x4* z = reinterpret_cast<u32x4*>(smem + Lds::fring + (kFN - 1) * kTileBytes) + tid * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) z[i] = u32x4{0u, 0u, 0u, 0u};
        if (tid < 64) reinterpret_cast<u32x4*>(smem + Lds::aring + (kFN - 1) * 1024)[tid] = u32x4{0u, 0u, 0u, 0u};
        reinterpret_cast<float2*>(smem + Lds::x2)[tid] = make_float2(0.f, 1.f);
    }
    {
        uint32_t off = 0;
        int px0 = y0 * W + x0, yrow = y0;
#pragma unroll
        for (int b = 0; b < kA; ++b) {
#pragma unroll
            for (int i = 0; i < 5; ++i) { if (i < nb) dma_piece(i, off, px0, yrow, b < nt); }
            off += kTileBytes;
            px0 += W;
            ++yrow;
        }
    }
    wait_vm<nb * (kA - 2)>();                                        // batches 0 and 1
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { convert_load(b * kTileBytes, i); convert_store(b * kTileBytes, i); }
    }
    wg_barrier();
#ifdef R4_STAMP
    const int wg_lin = blockIdx.y * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0 && wg_lin < 4096) {
        r4_clock[wg_lin][0] = __builtin_amdgcn_s_memtime();
        r4_clock[wg_lin][1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    {
        uint32_t off_l = 0, off_f = (kFN - 1) * kTileBytes, off_c = 2 * kTileBytes, off_d = kA * kTileBytes;
        int px0_d = (y0 + kA) * W + x0, yrow_d = y0 + kA;
        for (int it = 0; it <= nt; ++it) {
            {   // what tile it needs and no other wave writes in this iteration, read BEFORE the barrier: the accumulator's start value
                // Cy + Cx and the pixel's statistics (staged with the tile, landed iterations ago) ...
                const uint32_t cya = cy_lane + (off_l >> 4);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 a = lds4(cya + 32 * g), b = lds4(cx_lane + 32 * g);
#pragma unroll
                    for (int j = 0; j < 4; ++j) s[4 * g + j] = a[j] + b[j];
                }
                rt = *reinterpret_cast<SVPS_LDS const f32x2*>((uintptr_t)(rt_lane + (off_l >> 4)));
                asm volatile("" : "+v"(rt));
            }
            {   // ... and the first row fragments of tile it (its pieces became fp16 one iteration - or the prologue - ago): requested BEFORE the
                // barrier and still in flight behind it. LDS operations of a wave complete in order, so "all but the kFB - 1
                // youngest" covers every LDS write of the iteration (block statistics, converted pieces) that the barrier publishes.
                const uint32_t tb = lane_row + off_l;
                R4_FENCE();
#pragma unroll
                for (int f = 0; f < kFB - 1 - R4_KEEP; ++f) fb[f] = frag(tb, f);
                R4_FENCE();
            }
            R4_STAMP_AT(5);
#ifdef R4_STAMP
            wg_barrier();
#else
            asm volatile("s_waitcnt lgkmcnt(%0)\n\ts_barrier" ::"n"(kFB - 1 - R4_KEEP) : "memory");   // B(it)
#endif
            body(it, off_l, off_f, off_c, off_d, px0_d, yrow_d);
            off_f = off_l;
            off_l = ring_next(off_l);
            off_c = ring_next(off_c);
            off_d = ring_next(off_d);
            px0_d += W;
            ++yrow_d;
        }
    }
#ifdef R4_STAMP
    if (threadIdx.x == 0 && wg_lin < 4096) {
        r4_clock[wg_lin][2] = __builtin_amdgcn_s_memtime();
        r4_clock[wg_lin][3] = __builtin_amdgcn_s_memrealtime();
    }
    if (stamp_wg && lane < 64) (&r4_stamps[sb][0][0])[lane] = reinterpret_cast<const unsigned long long*>(smem + Lds::total)[sb * 64 + lane];
#endif
    // ---- partial sums of this chunk -> HBM ------------------------------------------------------------------------------------
    asm volatile("s_nop 15\n\ts_nop 2" ::: "memory");
    float* dst = A.partial + ((size_t)t * C + c) * A.L * kPartRow;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int slot = 32 * sb + acc_row(i, h);
        if (slot < A.L) {
#pragma unroll
            for (int db = 0; db < 8; ++db) dst[(size_t)slot * kPartRow + 32 * db + r] = o[db][i];
            if (r < 4) dst[(size_t)slot * kPartRow + 256 + r] = oa[i];     // aux columns 0 .. 2 (column 3 is zero); the rest is not data
        }
    }
}

__global__ __launch_bounds__(256) void retr_attn4_kernel(const Args A) {
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    switch (w) {                 // every role runs the same sequence of workgroup barriers
        case 0: role<0>(A); break;
        case 1: role<1>(A); break;
        case 2: role<2>(A); break;
        default: role<3>(A); break;
    }
}

// Sum of the C partials of every (frame, slot) row in chunk order (bitwise reproducible, no float atomics); out row (272 floats) =
// { A[0:256], s1, s0, 0 x 14 }: the operand of the slot-side product with [ (gamma_v W~_v)^T ; gamma_v b~_v ; beta_v ; 0 ].
__global__ __launch_bounds__(256) void retr_finish4_kernel(const float* __restrict__ partial, float* __restrict__ out, int L, int C) {
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

}  // namespace r4
}  // namespace svps

namespace {
struct Plan4 {
    int strips, cps, tpc;        // column strips, chunks per strip, tiles (image rows) per chunk
    int chunks() const { return strips * cps; }
};
// Chunks never leave a strip. All workgroups do nearly the same work, so the launch runs in rounds of one workgroup per CU: take the
// split whose last round is fullest, charging every workgroup its prologue (operands, table, ring fill: about five tile times).
Plan4 plan4(int T, int H, int W, int cps_req) {
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
            const double eff = (double)T * strips * H / ((double)rounds * cus * (tpc + 5));
            if (eff > best + 1e-9) { best = eff; cps = cc; }
        }
    }
    if (cps > H) cps = H;
    const int tpc = (H + cps - 1) / cps;
    cps = (H + tpc - 1) / tpc;
    return {strips, cps, tpc};
}
}  // namespace

extern "C" size_t svps_retr_attn4_workspace_bytes(int T, int L, int H, int W, int chunks_per_strip) {
    if (T <= 0 || L <= 0 || H <= 0 || W <= 0) return 0;
    const Plan4 p = plan4(T, H, W, chunks_per_strip);
    return (size_t)T * p.chunks() * L * svps::r4::kPartRow * sizeof(float);
}

extern "C" int svps_retr_attn4_fwd(const void* qh, const void* ql, const float* cy, const float* cx, const float* c3,
                                   const void* feat, const void* aux, void* workspace, size_t workspace_bytes, float* out_ext,
                                   int T, int L, int H, int W, int D, int chunks_per_strip, void* stream_) {
    if (!qh || !ql || !cy || !cx || !c3 || !feat || !aux || !workspace || !out_ext) return SVPS_ERR_BAD_ARG;
    if (D != svps::kD || T <= 0 || L <= 0 || L > 128 || H <= 0 || W <= 0) return SVPS_ERR_BAD_SHAPE;
    if ((size_t)H * W > svps::kMaxFramePixels) return SVPS_ERR_BAD_SHAPE;
    const Plan4 p = plan4(T, H, W, chunks_per_strip);
    const size_t need = (size_t)T * p.chunks() * L * svps::r4::kPartRow * sizeof(float);
    if (workspace_bytes < need) return SVPS_ERR_WORKSPACE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    svps::r4::Args a;
    a.qh = static_cast<const _Float16*>(qh);
    a.ql = static_cast<const _Float16*>(ql);
    a.cy = cy; a.cx = cx; a.c3g = c3;
    a.feat = static_cast<const __bf16*>(feat);
    a.aux = static_cast<const __bf16*>(aux);
    a.partial = static_cast<float*>(workspace);
    a.L = L; a.HW = H * W; a.H = H; a.W = W; a.tiles_per_chunk = p.tpc; a.chunks_per_strip = p.cps;
    static SvpsLdsAttr attr;
#ifdef R4_STAMP
    constexpr int kLdsBytes = svps::r4::Lds::total + 2048;      // + the stamps of the diagnostic build
#else
    constexpr int kLdsBytes = svps::r4::Lds::total;
#endif
    if (hipError_t ae = attr.ensure(reinterpret_cast<const void*>(svps::r4::retr_attn4_kernel), kLdsBytes); ae != hipSuccess) return (int)ae;
    svps_prof_mark(SVPS_KERNEL_RETR_ATTN, 0, stream);
    hipLaunchKernelGGL(svps::r4::retr_attn4_kernel, dim3(p.chunks(), T), dim3(256), kLdsBytes, stream, a);
    hipError_t e = hipGetLastError();
    svps_prof_mark(SVPS_KERNEL_RETR_ATTN, 1, stream);
    if (e != hipSuccess) return (int)e;
    svps_prof_mark(SVPS_KERNEL_RETR_FINISH, 0, stream);
    hipLaunchKernelGGL(svps::r4::retr_finish4_kernel, dim3(L, T), dim3(256), 0, stream, static_cast<const float*>(workspace), out_ext, L, p.chunks());
    svps_prof_mark(SVPS_KERNEL_RETR_FINISH, 1, stream);
    return (int)hipGetLastError();
}

#ifdef R4_STAMP
extern "C" int svps_retr4_debug_read(unsigned long long* stamps, unsigned long long* clock) {
    hipError_t e = hipMemcpyFromSymbol(stamps, HIP_SYMBOL(svps::r4::r4_stamps), sizeof(unsigned long long) * 4 * 8 * 8);
    if (e != hipSuccess) return (int)e;
    return (int)hipMemcpyFromSymbol(clock, HIP_SYMBOL(svps::r4::r4_clock), sizeof(unsigned long long) * 4096 * 4);
}
#endif
