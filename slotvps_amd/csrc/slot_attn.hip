// K1 - slot <-> pixel retriever ("Panoptic Retriever" cross-attention) for gfx950.
//
// Replaces the body of MaskDynamicConv.forward in the reference
// (mmdet/models/detectors/dynamic_mask_head.py:435-459) after the q/k/v projections + LayerNorms:
//
//     A[l, p]   = sum_c q[l, c] * k[p, c]                  (unscaled logits,          :435)
//     P[:, p]   = softmax over the SLOT axis l, per pixel  (F.softmax(dim=1),         :446)
//     o[l, :]   = sum_p P[l, p] * v[p, :]                  (plain sum over pixels,    :456)
//     out[l, :] = ReLU(LayerNorm(o[l, :]))                 (norm1 + activation,       :458-459)
//
// Because the softmax runs over slots, every pixel column is independent: no online-softmax
// rescaling across pixel tiles is needed and the only cross-tile state is the [L, 256] fp32 sum.
//
// Mapping onto the CU
//   * one workgroup = NW waves (4 for L <= 128, 8 for L <= 256); wave w owns slots [32w, 32w+32)
//     for the whole kernel: its 32 x 256 query block lives in 64 VGPRs as MFMA A fragments and its
//     32 x 256 fp32 output block in 128 accumulator registers.
//   * the workgroup walks a contiguous range of 32-pixel tiles of ONE frame. k and v tiles
//     (16 KiB each, bf16 [pixel][channel]) arrive by LDS-DMA into a ring of NST stages, issued
//     NST-1 tiles ahead behind a counted vmcnt and a raw s_barrier (the DMA is never drained in
//     the loop). Each HBM byte of k and v is read exactly once.
//   * S = q k^T: 16 MFMA 32x32x16 per tile and wave, keys read from LDS as row fragments
//     (ds_read_b128, XOR swizzle). Pixel = accumulator column = lane, so the softmax over this
//     wave's 32 slots is an in-lane reduction + one lane^32 exchange; the 4 (8) waves then trade
//     one (max, sum) pair per pixel through LDS: ONE barrier per tile for the softmax.
//   * P is rounded to bf16 (optionally hi + lo for a 16-bit mantissa, flag SVPS_FLAG_SPLIT_P),
//     transposed through a wave-private LDS image and multiplied with v fragments obtained with
//     ds_read_b64_tr_b16 (hardware transpose, pixel-major v needs no transposed copy in HBM).
//   * every workgroup stores its [L, 256] fp32 partial; `slot_attn_finish` sums the partials in a
//     fixed order (bitwise reproducible, no float atomics), applies LayerNorm + ReLU.
#include "common.h"

namespace svps {

constexpr int kPimgStride = 80;  // bytes per slot row of the P image: 32 px * 2 B + 16 B pad

template <int NW, int NST>
struct AttnLds {
    static constexpr int ring = 0;                                   // NST * (k tile + v tile)
    static constexpr int stats = NST * 2 * kTileBytes;               // NW * 32 * float2
    static constexpr int pimg = stats + NW * kTilePx * 8;            // NW * 2 * 32 * 80 B
    static constexpr int total = pimg + NW * 2 * 32 * kPimgStride;
};

template <int NW, int NST, bool SPLIT>
__global__ __launch_bounds__(NW * 64) void slot_attn_partial(
    const __bf16* __restrict__ q,   // [T, L, 256]
    const __bf16* __restrict__ k,   // [T, HW, 256]
    const __bf16* __restrict__ v,   // [T, HW, 256]
    float* __restrict__ partial,    // [T, C, L, 256]
    int L, int HW, int tiles_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using Lds = AttnLds<NW, NST>;

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int t = blockIdx.y, c = blockIdx.x, C = gridDim.x;

    const int px_begin = c * tiles_per_chunk * kTilePx;
    int px_end = px_begin + tiles_per_chunk * kTilePx;
    px_end = px_end < HW ? px_end : HW;
    const int nt = (px_end - px_begin + kTilePx - 1) / kTilePx;  // >= 1 by construction

    const char* kb = reinterpret_cast<const char*>(k) + (size_t)t * HW * kRowBytes;
    const char* vb = reinterpret_cast<const char*>(v) + (size_t)t * HW * kRowBytes;

    // ---- query block of this wave -> registers (A fragments), zero rows past L ------------
    bf16x8 qf[16];
    {
        const int slot = 32 * w + r;
        const __bf16* qrow = q + ((size_t)t * L + (slot < L ? slot : 0)) * kD + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            u32x4 raw = *reinterpret_cast<const u32x4*>(qrow + 16 * ks);
            if (slot >= L) raw = u32x4{0u, 0u, 0u, 0u};
            qf[ks] = __builtin_bit_cast(bf16x8, raw);
        }
    }
    // Make sure the query loads have landed before LDS-DMA starts, so that the compiler's wait for
    // them does not sit behind (and drain) the DMA ring later on.
    wait_vm<0>();

    // ---- prologue: NST-1 tiles in flight ------------------------------------------------------
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) {
        if (s < nt) {
            char* st = smem + Lds::ring + s * 2 * kTileBytes;
            dma_tile<NW>(kb, px_begin + s * kTilePx, HW - 1, st, w, lane);
            dma_tile<NW>(vb, px_begin + s * kTilePx, HW - 1, st + kTileBytes, w, lane);
        }
    }

    f32x16 o[8];
#pragma unroll
    for (int db = 0; db < 8; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] = 0.f;

    float2* stats = reinterpret_cast<float2*>(smem + Lds::stats);
    char* pimg_hi = smem + Lds::pimg + w * 2 * 32 * kPimgStride;
    char* pimg_lo = pimg_hi + 32 * kPimgStride;
    constexpr int PIECES = 2 * (kTilePx / 2 / NW);  // DMA pieces per wave and tile (k + v)

    for (int it = 0; it < nt; ++it) {
        // -- tile `it` has landed for this wave; the barrier extends that to every wave and also
        //    says every wave is done reading tile it-1 (whose stage is refilled just below).
        if constexpr (NST == 4) {
            if (it + 2 < nt) wait_vm<2 * PIECES>();
            else if (it + 1 < nt) wait_vm<PIECES>();
            else wait_vm<0>();
        } else if constexpr (NST == 3) {
            if (it + 1 < nt) wait_vm<PIECES>();
            else wait_vm<0>();
        } else {
            wait_vm<0>();
        }
        wg_barrier();
        if (it + NST - 1 < nt) {
            char* st = smem + Lds::ring + ((it + NST - 1) % NST) * 2 * kTileBytes;
            const int px0 = px_begin + (it + NST - 1) * kTilePx;
            dma_tile<NW>(kb, px0, HW - 1, st, w, lane);
            dma_tile<NW>(vb, px0, HW - 1, st + kTileBytes, w, lane);
        }
        const char* kt = smem + Lds::ring + (it % NST) * 2 * kTileBytes;
        const char* vt = kt + kTileBytes;

        // -- logits of this wave's 32 slots x 32 pixels ---------------------------------------
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[ks], read_row_frag(kt, ks, r, h), s, 0, 0, 0);

        // -- softmax over slots: local (max, sum) for pixel column r ---------------------------
        const int slot0 = 32 * w + 4 * h;
        float mloc = kNegBig;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bool ok = slot0 + (i & 3) + 8 * (i >> 2) < L;
            s[i] = ok ? s[i] : kNegBig;
            mloc = fmaxf(mloc, s[i]);
        }
        mloc = wave_half_xor_max(mloc);
        float sloc = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bool ok = slot0 + (i & 3) + 8 * (i >> 2) < L;
            const float e = __builtin_amdgcn_exp2f((s[i] - mloc) * kLog2e);
            s[i] = ok ? e : 0.f;
            sloc += s[i];
        }
        sloc = wave_half_xor_sum(sloc);
        if (h == 0) stats[w * kTilePx + r] = make_float2(mloc, sloc);
        wg_barrier();
        float mall = kNegBig;
        float2 st_w[NW];
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) {
            st_w[ww] = stats[ww * kTilePx + r];
            mall = fmaxf(mall, st_w[ww].x);
        }
        float den = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww)
            den += st_w[ww].y * __builtin_amdgcn_exp2f((st_w[ww].x - mall) * kLog2e);
        float fac = __builtin_amdgcn_exp2f((mloc - mall) * kLog2e) / den;
        if (px_begin + it * kTilePx + r >= px_end) fac = 0.f;  // pixels past the chunk / frame

        // -- P -> bf16 (hi [+ lo]) -> wave-private [slot][pixel] image --------------------------
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float p = s[i] * fac;
            const __bf16 ph = (__bf16)p;
            const int off = acc_row(i, h) * kPimgStride + r * 2;
            *reinterpret_cast<__bf16*>(pimg_hi + off) = ph;
            if constexpr (SPLIT) *reinterpret_cast<__bf16*>(pimg_lo + off) = (__bf16)(p - (float)ph);
        }
        wait_lgkm0();

        // -- o += P v ------------------------------------------------------------------------
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int aoff = r * kPimgStride + 32 * ks + 16 * h;
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(pimg_hi + aoff);
            bf16x8 al;
            if constexpr (SPLIT) al = *reinterpret_cast<const bf16x8*>(pimg_lo + aoff);
#pragma unroll
            for (int db = 0; db < 8; ++db) {
                const bf16x8 vf = read_col_frag(vt, ks, db, lane);
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, vf, o[db], 0, 0, 0);
                if constexpr (SPLIT) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, vf, o[db], 0, 0, 0);
            }
        }
    }

    // ---- partial [L, 256] of this workgroup -----------------------------------------------------
    float* dst = partial + ((size_t)t * C + c) * L * kD;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int slot = 32 * w + acc_row(i, h);
        if (slot < L) {
#pragma unroll
            for (int db = 0; db < 8; ++db) dst[(size_t)slot * kD + 32 * db + r] = o[db][i];
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Wave-specialised variant for L <= 128: 8 waves = 4 producers + 4 consumers, two waves per SIMD.
//
// The 4-wave kernel above serialises  logits -> softmax -> P -> PV  inside every wave, so the
// matrix pipe idles during the softmax and the VALU idles during the MFMAs. Here the two halves
// run on DIFFERENT waves of the same SIMD, one 32-pixel tile apart:
//   producer wave sb (waves 0-3): logits of slot block sb for tile i (16 MFMA), softmax over slots
//       (in-lane + lane^32 + one (max, sum) exchange with the other three producers), P(i) -> LDS
//   consumer wave sb (waves 4-7): o[sb, all 256 channels] += P(i-1) v(i-1) (16 MFMA, 32 with
//       SPLIT) and ALL the LDS-DMA of the workgroup (the producers never touch vector memory)
// so a SIMD's matrix pipe (mostly the consumer) and its VALU (the producer) are busy together.
// Two barriers per tile; the consumers cross the softmax barrier between their two 16-pixel
// k-steps. (A three-stage variant with one barrier per tile measured 5-7 % slower.)
//
// LDS: keys in a 5-deep ring, values in a 4-deep ring (16 KiB tiles, 144 KiB), fetched 3 tiles
// ahead. P(i) is written over k(i) - the keys are dead once all four producers have their logits -
// as [slot block][pixel][slot] (8-byte packed stores, read back as MFMA A fragments with
// ds_read_b64_tr_b16). k / v tiles arrive by `buffer_load ... lds` with tile-invariant per-lane
// offsets plus a scalar tile offset.
// ------------------------------------------------------------------------------------------------
constexpr int kPrefetch = 3;             // tiles in flight ahead of the one being consumed
constexpr int kNK = kPrefetch + 2;       // key / P ring depth
constexpr int kNV = kPrefetch + 1;       // value ring depth

struct AttnWsLds {
    static constexpr int kring = 0;
    static constexpr int vring = kNK * kTileBytes;
    static constexpr int stats = vring + kNV * kTileBytes;   // [4][32] float2
    static constexpr int total = stats + 4 * 32 * 8;
};

// Raw buffer descriptor (4 SGPRs) for `bytes` bytes at `base`; every word is made provably
// wave-uniform so the asm below can take it in scalar registers.
__device__ __forceinline__ u32x4 make_srd(const void* base, uint32_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    u32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
    d[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);   // stride = 0
    d[2] = __builtin_amdgcn_readfirstlane(bytes);
    d[3] = 0x00020000u;
    return d;
}

// One 1-KiB LDS-DMA piece, `buffer_load_dwordx4 ... offen lds`: LDS[lds_addr + lane*16] <-
// buffer[soff + voff(lane)]. Written in asm on purpose: hipcc treats the builtin form as an LDS
// store that may alias every later ds_read and drains it with s_waitcnt vmcnt(0) before the first
// LDS read that follows - which serialises the whole prefetch ring. In asm the load is invisible to
// the compiler's counters; the kernel orders it by hand (counted vmcnt, then a workgroup barrier,
// then the reads). M0 (the LDS base of the DMA) is saved and restored inside the statement.
__device__ __forceinline__ void dma16_srd(u32x4 srd, uint32_t lds_addr, int voff, int soff) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        // nt: k and v are read exactly once - non-temporal loads land ~18 % sooner (DMA-only ablation of
        // this kernel: 118 us -> 104 us per 671 MB launch = 6.4 TB/s)
        "buffer_load_dwordx4 %2, %3, %4 offen nt lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_addr), "v"(voff), "s"(srd), "s"(soff)
        : "memory");
}

__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return (uint32_t)reinterpret_cast<uintptr_t>((SVPS_LDS const void*)p);
}

__device__ __forceinline__ float half_swap_max(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_swap_sum(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// ABL != 0 are timing-only ablations for tools/kbench.py (env SVPS_ABLATE); their outputs are wrong.
//   1: DMA + barriers only   2: producers only (no PV)   4: consumers only (no logits / softmax)   5: no DMA
// EXT (more than 128 slots): the launch covers the `L` slots starting at row `slot_off` of a query / partial layout with
// `Lrow` rows per frame, and the per-pixel softmax statistics over ALL slots come from `ext_stats` ([T, HW] of
// (max logit, 1 / sum of exponentials), written by slot_attn_stats) instead of the exchange between the four producers.
template <bool SPLIT, int ABL = 0, bool EXT = false>
__global__ __launch_bounds__(512) void slot_attn_partial_ws(
    const __bf16* __restrict__ q,   // [T, Lrow, 256]
    const __bf16* __restrict__ k,   // [T, HW, 256]
    const __bf16* __restrict__ v,   // [T, HW, 256]
    float* __restrict__ partial,    // [T, C, Lrow, 256]
    int L, int HW, int tiles_per_chunk, int Lrow, int slot_off, const float2* __restrict__ ext_stats) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using Lds = AttnWsLds;
    constexpr int A = kPrefetch;

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sb = w & 3;
    const bool consumer = w >= 4;
    const int r = lane & 31, h = lane >> 5;
    const int t = blockIdx.y, c = blockIdx.x, C = gridDim.x;

    const int px_begin = c * tiles_per_chunk * kTilePx;
    int px_end = px_begin + tiles_per_chunk * kTilePx;
    px_end = px_end < HW ? px_end : HW;
    const int nt = (px_end - px_begin + kTilePx - 1) / kTilePx;

    float2* stats = reinterpret_cast<float2*>(smem + Lds::stats);

    if (!consumer) {
        // ================================ producer =============================================
        bf16x8 qf[16];
        {
            const int slot = 32 * sb + r;
            const __bf16* qrow = q + ((size_t)t * Lrow + slot_off + (slot < L ? slot : 0)) * kD + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                u32x4 raw = *reinterpret_cast<const u32x4*>(qrow + 16 * ks);
                if (slot >= L) raw = u32x4{0u, 0u, 0u, 0u};
                qf[ks] = __builtin_bit_cast(bf16x8, raw);
            }
        }
        const int slot0 = 32 * sb + 4 * h;
        const int key = (r >> 1) & 3;
        float2 ext_cur = make_float2(0.f, 0.f);
        if constexpr (EXT) {
            const int px = px_begin + r;
            ext_cur = ext_stats[(size_t)t * HW + (px < HW ? px : HW - 1)];
        }
        for (int it = 0; it <= nt; ++it) {
            wg_barrier();                                        // B_top(it)
            if (it == nt || ABL == 1 || ABL == 4) { wg_barrier(); continue; }
            char* kt = smem + Lds::kring + (it % kNK) * kTileBytes;
            // all 16 key fragments are requested before the first MFMA (hipcc otherwise keeps only
            // two LDS reads in flight and the dependent MFMA chain runs at LDS latency)
            bf16x8 kf[16];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) kf[ks] = read_row_frag(kt, ks, r, h);
            __builtin_amdgcn_sched_barrier(0);
            f32x16 s;
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[ks], kf[ks], s, 0, 0, 0);
            if constexpr (EXT) {
                // statistics of this tile's pixels are in registers (prefetched one tile ahead): no exchange, one pass
                const float2 st = ext_cur;
                {
                    int px = px_begin + (it + 1) * kTilePx + r;
                    px = px < HW ? px : HW - 1;
                    ext_cur = ext_stats[(size_t)t * HW + px];           // next tile; consumed after two barriers
                }
                const float mneg = -st.x * kLog2e;
                float fac = st.y;
                if (px_begin + it * kTilePx + r >= px_end) fac = 0.f;   // pixels past the chunk / frame
                wg_barrier();                                        // B_stats(it): "k(it) is dead"
                char* prow = kt + sb * 2048 + r * 64 + 8 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 ph, pl;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool ok = slot0 + j + 8 * g < L;
                        const float p = ok ? __builtin_amdgcn_exp2f(fmaf(s[4 * g + j], kLog2e, mneg)) * fac : 0.f;
                        ph[j] = (__bf16)p;
                        if constexpr (SPLIT) pl[j] = (__bf16)(p - (float)ph[j]);
                    }
                    *reinterpret_cast<bf16x4*>(prow + ((g ^ key) * 16)) = ph;
                    if constexpr (SPLIT) *reinterpret_cast<bf16x4*>(prow + 8192 + ((g ^ key) * 16)) = pl;
                }
                continue;
            }
            float mloc = kNegBig;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const bool ok = slot0 + (i & 3) + 8 * (i >> 2) < L;
                s[i] = ok ? s[i] : kNegBig;
                mloc = fmaxf(mloc, s[i]);
            }
            mloc = half_swap_max(mloc);
            float sloc = 0.f;
            const float mneg = -mloc * kLog2e;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const bool ok = slot0 + (i & 3) + 8 * (i >> 2) < L;
                const float e = __builtin_amdgcn_exp2f(fmaf(s[i], kLog2e, mneg));
                s[i] = ok ? e : 0.f;
                sloc += s[i];
            }
            sloc = half_swap_sum(sloc);
            if (h == 0) stats[sb * 32 + r] = make_float2(mloc, sloc);
            wg_barrier();                                        // B_stats(it): also "k(it) is dead"
            float mall = kNegBig;
            float2 st_w[4];
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) {
                st_w[ww] = stats[ww * 32 + r];
                mall = fmaxf(mall, st_w[ww].x);
            }
            float den = 0.f;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww)
                den += st_w[ww].y * __builtin_amdgcn_exp2f((st_w[ww].x - mall) * kLog2e);
            float fac = __builtin_amdgcn_exp2f((mloc - mall) * kLog2e) / den;
            if (px_begin + it * kTilePx + r >= px_end) fac = 0.f;   // pixels past the chunk / frame
            char* prow = kt + sb * 2048 + r * 64 + 8 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 ph, pl;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float p = s[4 * g + j] * fac;
                    ph[j] = (__bf16)p;
                    if constexpr (SPLIT) pl[j] = (__bf16)(p - (float)ph[j]);
                }
                *reinterpret_cast<bf16x4*>(prow + ((g ^ key) * 16)) = ph;
                if constexpr (SPLIT) *reinterpret_cast<bf16x4*>(prow + 8192 + ((g ^ key) * 16)) = pl;
            }
        }
        return;
    }

    // =================================== consumer ===============================================
    const u32x4 krs = make_srd(k + (size_t)t * HW * kD, (uint32_t)HW * kRowBytes);
    const u32x4 vrs = make_srd(v + (size_t)t * HW * kD, (uint32_t)HW * kRowBytes);
    // consumer wave sb stages rows 8 sb .. 8 sb + 7 of every k and v tile (4 pieces of 2 rows each)
    int voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * sb + 2 * i + h;
        voff[i] = row * kRowBytes + (((lane & 31) ^ swz(row)) * 16);
    }
    const uint32_t lds0 = lds_addr_of(smem);
    auto stage = [&](u32x4 rs, int tile, int buf_off) {
        const uint32_t st = __builtin_amdgcn_readfirstlane(lds0 + buf_off + sb * 4 * 1024);
        const int px0 = px_begin + tile * kTilePx;
        const int soff = __builtin_amdgcn_readfirstlane(px0 * kRowBytes);
        if (px0 + kTilePx <= HW) {
#pragma unroll
            for (int i = 0; i < 4; ++i) dma16_srd(rs, st + i * 1024, voff[i], soff);
        } else {  // ragged last tile of the frame: clamp source rows (their P is forced to 0)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 8 * sb + 2 * i + h;
                const int src = px0 + row < HW ? row : HW - 1 - px0;
                dma16_srd(rs, st + i * 1024, src * kRowBytes + (((lane & 31) ^ swz(row)) * 16), soff);
            }
        }
    };
    // batch b = { k(b), v(b-1) } must have landed by B_top(b); batches 0 .. A-1 go out here, batch
    // it + A right after B_top(it)
    auto issue_batch = [&](int b) {
        if constexpr (ABL == 5) { if (b > 1) return; }
        if (b < nt) stage(krs, b, Lds::kring + (b % kNK) * kTileBytes);
        if (b >= 1 && b - 1 < nt) stage(vrs, b - 1, Lds::vring + ((b - 1) % kNV) * kTileBytes);
    };
#pragma unroll
    for (int b = 0; b < A; ++b) issue_batch(b);

    f32x16 o[8];
#pragma unroll
    for (int db = 0; db < 8; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] = 0.f;

    const int g2 = lane >> 4, ii = lane & 15, qq = ii >> 2, pp = ii & 3;
    const int pchunk = 2 * (g2 & 1) + (pp >> 1);

    auto pv_step = [&](const char* pt, const char* vt, int ks) {
        const int px0 = 16 * ks + 8 * (g2 >> 1) + qq;
        const char* a0 = pt + sb * 2048 + 8 * (pp & 1) + px0 * 64 + ((pchunk ^ ((px0 >> 1) & 3)) * 16);
        const char* a1 = pt + sb * 2048 + 8 * (pp & 1) + (px0 + 4) * 64 + ((pchunk ^ (((px0 + 4) >> 1) & 3)) * 16);
        const bf16x8 ah = __builtin_shufflevector(
            __builtin_amdgcn_ds_read_tr16_b64_v4bf16((SVPS_LDS bf16x4*)a0),
            __builtin_amdgcn_ds_read_tr16_b64_v4bf16((SVPS_LDS bf16x4*)a1), 0, 1, 2, 3, 4, 5, 6, 7);
        bf16x8 al;
        if constexpr (SPLIT)
            al = __builtin_shufflevector(
                __builtin_amdgcn_ds_read_tr16_b64_v4bf16((SVPS_LDS bf16x4*)(a0 + 8192)),
                __builtin_amdgcn_ds_read_tr16_b64_v4bf16((SVPS_LDS bf16x4*)(a1 + 8192)), 0, 1, 2, 3, 4, 5, 6, 7);
        bf16x8 vf[8];
#pragma unroll
        for (int db = 0; db < 8; ++db) vf[db] = read_col_frag(vt, ks, db, lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int db = 0; db < 8; ++db) {
            o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, vf[db], o[db], 0, 0, 0);
            if constexpr (SPLIT) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, vf[db], o[db], 0, 0, 0);
        }
    };

    for (int it = 0; it <= nt; ++it) {
        // batch `it` landed for this wave: all but the A-1 younger batches (8 loads each; the tail
        // batches are smaller, so the last iterations simply drain)
        if (it + A - 1 < nt) wait_vm<8 * (A - 1)>();
        else wait_vm<0>();
        wg_barrier();                                            // B_top(it)
        if (it + A <= nt) issue_batch(it + A);
        const bool work = it >= 1 && ABL != 1 && ABL != 2;
        const char* pt = smem + Lds::kring + ((it + kNK - 1) % kNK) * kTileBytes;   // P(it-1) over k(it-1)
        const char* vt = smem + Lds::vring + ((it + kNV - 1) % kNV) * kTileBytes;   // v(it-1)
        if (work) pv_step(pt, vt, 0);
        wg_barrier();                                            // B_stats(it)
        if (work) pv_step(pt, vt, 1);
    }

    float* dst = partial + (((size_t)t * C + c) * Lrow + slot_off) * kD;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int slot = 32 * sb + acc_row(i, h);
        if (slot < L) {
#pragma unroll
            for (int db = 0; db < 8; ++db) dst[(size_t)slot * kD + 32 * db + r] = o[db][i];
        }
    }
}

// Per-pixel softmax statistics over up to 256 slots (first kernel of the two-kernel path for more than 128 slots):
// stats[t, p] = (max_l S[l, p], 1 / sum_l exp(S[l, p] - max)), S = q k^T. Eight waves, wave w owns slots [32w, 32w + 32);
// keys stream through a 4-deep asm LDS-DMA ring (two 1-KiB pieces per wave and tile); reads k once, writes 8 B per pixel.
struct StatsLds {
    static constexpr int kStages = 4;
    static constexpr int ring = 0;
    static constexpr int stats = kStages * kTileBytes;          // [8][32] float2
    static constexpr int total = stats + 8 * 32 * 8;
};

__global__ __launch_bounds__(512) void slot_attn_stats(const __bf16* __restrict__ q,   // [T, L, 256]
                                                       const __bf16* __restrict__ k,   // [T, HW, 256]
                                                       float2* __restrict__ out,       // [T, HW]
                                                       int L, int HW, int tiles_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using Lds = StatsLds;
    constexpr int NST = Lds::kStages, A = NST - 1;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int t = blockIdx.y, c = blockIdx.x;
    const int px_begin = c * tiles_per_chunk * kTilePx;
    int px_end = px_begin + tiles_per_chunk * kTilePx;
    px_end = px_end < HW ? px_end : HW;
    const int nt = (px_end - px_begin + kTilePx - 1) / kTilePx;

    bf16x8 qf[16];
    {
        const int slot = 32 * w + r;
        const __bf16* qrow = q + ((size_t)t * L + (slot < L ? slot : 0)) * kD + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            u32x4 raw = *reinterpret_cast<const u32x4*>(qrow + 16 * ks);
            if (slot >= L) raw = u32x4{0u, 0u, 0u, 0u};
            qf[ks] = __builtin_bit_cast(bf16x8, raw);
        }
    }
    wait_vm<0>();
    const u32x4 krs = make_srd(k + (size_t)t * HW * kD, (uint32_t)HW * kRowBytes);
    const u32x4 ors = make_srd(out + (size_t)t * HW, (uint32_t)HW * 8u);
    const uint32_t lds0 = lds_addr_of(smem);
    auto stage = [&](int tile) {                                     // rows 4w .. 4w + 3 of the tile: two pieces
        if (tile >= nt) return;
        const int px0 = px_begin + tile * kTilePx;
        const int soff = __builtin_amdgcn_readfirstlane(px0 * kRowBytes);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 4 * w + 2 * i + h;
            const int src = px0 + row < HW ? row : HW - 1 - px0;     // ragged last tile: clamp (those pixels are not stored)
            const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + Lds::ring + (tile % NST) * kTileBytes + (4 * w + 2 * i) * kRowBytes);
            dma16_srd(krs, dst, src * kRowBytes + (((lane & 31) ^ swz(row)) * 16), soff);
        }
    };
#pragma unroll
    for (int b = 0; b < A; ++b) stage(b);
    float2* stats = reinterpret_cast<float2*>(smem + Lds::stats);
    const int slot0 = 32 * w + 4 * h;
    for (int it = 0; it < nt; ++it) {
        // k(it) landed for this wave. Younger, in issue order: store(it-3), DMA(it+1), store(it-2), DMA(it+2), store(it-1)
        wait_vm_dyn((it < 3 ? it : 3) + 2 * ((it + 1 < nt) + (it + 2 < nt)));
        wg_barrier();                                                // B_top(it); also: every wave is done with k(it-1)
        stage(it + A);
        const char* kt = smem + Lds::ring + (it % NST) * kTileBytes;
        bf16x8 kf[16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) kf[ks] = read_row_frag(kt, ks, r, h);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[ks], kf[ks], s, 0, 0, 0);
        float mloc = kNegBig;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bool ok = slot0 + (i & 3) + 8 * (i >> 2) < L;
            s[i] = ok ? s[i] : kNegBig;
            mloc = fmaxf(mloc, s[i]);
        }
        mloc = half_swap_max(mloc);
        float sloc = 0.f;
        const float mneg = -mloc * kLog2e;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bool ok = slot0 + (i & 3) + 8 * (i >> 2) < L;
            sloc += ok ? __builtin_amdgcn_exp2f(fmaf(s[i], kLog2e, mneg)) : 0.f;
        }
        sloc = half_swap_sum(sloc);
        if (h == 0) stats[w * 32 + r] = make_float2(mloc, sloc);
        wg_barrier();                                                // B_stats(it)
        // wave w combines and stores pixels 4w .. 4w + 3 (one store instruction per wave and tile, always issued)
        float mall = kNegBig;
        float2 st_w[8];
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) {
            st_w[ww] = stats[ww * 32 + r];
            mall = fmaxf(mall, st_w[ww].x);
        }
        float den = 0.f;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) den += st_w[ww].y * __builtin_amdgcn_exp2f((st_w[ww].x - mall) * kLog2e);
        const int px = px_begin + it * kTilePx + r;
        const bool mine = h == 0 && (r >> 2) == w && px < px_end;
        typedef __attribute__((ext_vector_type(2))) float f32x2v;
        const f32x2v val = {mall, 1.f / den};
        const int voff = mine ? px * 8 : 0x7ffffff0;                 // out of range -> dropped by the hardware range check
        asm volatile("buffer_store_dwordx2 %0, %1, %2, 0 offen" : : "v"(val), "v"(voff), "s"(ors) : "memory");
    }
}

// Sum the C partials of every (frame, slot) row in chunk order, then LayerNorm (biased variance,
// two-pass, eps inside the sqrt - torch.nn.LayerNorm semantics) and ReLU. One 256-thread
// workgroup per row, thread = channel.
__global__ __launch_bounds__(256) void slot_attn_finish(const float* __restrict__ partial,
                                                        const float* __restrict__ ln_w,
                                                        const float* __restrict__ ln_b, float eps,
                                                        float* __restrict__ out,      // [T, L, 256]
                                                        float* __restrict__ out_pre,  // or null
                                                        int L, int C) {
    __shared__ float red[8];
    const int l = blockIdx.x, t = blockIdx.y, d = threadIdx.x;
    const float* src = partial + ((size_t)t * C * L + l) * kD + d;
    // fixed summation order (bitwise reproducible): 8 interleaved partial sums keep 8 loads in
    // flight per thread, then a fixed-order combine
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const size_t cstride = (size_t)L * kD;
    int c = 0;
    for (; c + 8 <= C; c += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) a8[u] += src[(size_t)(c + u) * cstride];
    }
    for (int u = 0; c < C; ++c, ++u) a8[u] += src[(size_t)c * cstride];
    float acc = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    if (out_pre) out_pre[((size_t)t * L + l) * kD + d] = acc;

    auto block_sum = [&](float x) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m);
        __syncthreads();
        if ((d & 63) == 0) red[d >> 6] = x;
        __syncthreads();
        return red[0] + red[1] + red[2] + red[3];
    };
    const float mean = block_sum(acc) * (1.f / kD);
    const float dev = acc - mean;
    const float var = block_sum(dev * dev) * (1.f / kD);
    float y = dev * rsqrtf(var + eps) * ln_w[d] + ln_b[d];
    out[((size_t)t * L + l) * kD + d] = y > 0.f ? y : 0.f;
}

}  // namespace svps

// ------------------------------------------------------------------------------------------------
// C ABI (declared in include/slotvps_hip.h)
// ------------------------------------------------------------------------------------------------
#include <stdlib.h>

#include "../../include/slotvps_hip.h"

namespace {

int num_cus() { return svps_num_cus(); }

struct AttnPlan {
    int chunks;           // workgroups per frame
    int tiles_per_chunk;  // pixel tiles per workgroup
    int tile_px;          // pixels per tile (32)
};

AttnPlan plan_attn(int T, int L, int HW, int chunks_req) {
    const int tile_px = svps::kTilePx;
    const int tiles = (HW + tile_px - 1) / tile_px;
    int chunks = chunks_req;
    if (chunks <= 0) {
        // one resident workgroup per CU: the launch is a single wave of workgroups, each streaming
        // one contiguous pixel range of one frame
        chunks = svps_pick_chunks(T, tiles, num_cus(), 64);   // every workgroup costs a [L, 256] fp32 partial and a query load
    }
    if (chunks > tiles) chunks = tiles;
    int tpc = (tiles + chunks - 1) / chunks;
    chunks = (tiles + tpc - 1) / tpc;  // drop empty trailing chunks
    return {chunks, tpc, tile_px};
}

template <int NW, int NST, bool SPLIT>
hipError_t launch_partial(const void* q, const void* k, const void* v, float* partial, int T, int L,
                          int HW, const AttnPlan& p, hipStream_t stream) {
    using Lds = svps::AttnLds<NW, NST>;
    auto kern = svps::slot_attn_partial<NW, NST, SPLIT>;
    static SvpsLdsAttr attr;
    if (hipError_t ae = attr.ensure(reinterpret_cast<const void*>(kern), Lds::total); ae != hipSuccess) return ae;
    hipLaunchKernelGGL(kern, dim3(p.chunks, T), dim3(NW * 64), Lds::total, stream,
                       static_cast<const __bf16*>(q), static_cast<const __bf16*>(k),
                       static_cast<const __bf16*>(v), partial, L, HW, p.tiles_per_chunk);
    return hipGetLastError();
}

template <bool SPLIT, int ABL = 0, bool EXT = false>
hipError_t launch_partial_ws(const void* q, const void* k, const void* v, float* partial, int T, int L,
                             int HW, const AttnPlan& p, hipStream_t stream, int Lrow = 0, int slot_off = 0,
                             const float2* ext_stats = nullptr) {
    using Lds = svps::AttnWsLds;
    auto kern = svps::slot_attn_partial_ws<SPLIT, ABL, EXT>;
    if (Lrow == 0) Lrow = L;
    static SvpsLdsAttr attr;
    if (hipError_t ae = attr.ensure(reinterpret_cast<const void*>(kern), Lds::total); ae != hipSuccess) return ae;
    hipLaunchKernelGGL(kern, dim3(p.chunks, T), dim3(512), Lds::total, stream,
                       static_cast<const __bf16*>(q), static_cast<const __bf16*>(k),
                       static_cast<const __bf16*>(v), partial, L, HW, p.tiles_per_chunk, Lrow, slot_off, ext_stats);
    return hipGetLastError();
}

// more than 128 slots: statistics kernel, then the specialised kernel once per half of the slots
template <bool SPLIT>
hipError_t launch_two_pass(const void* q, const void* k, const void* v, float* partial, float2* stats, int T, int L,
                           int HW, const AttnPlan& p, hipStream_t stream) {
    static SvpsLdsAttr attr;
    if (hipError_t ae = attr.ensure(reinterpret_cast<const void*>(svps::slot_attn_stats), svps::StatsLds::total); ae != hipSuccess) return ae;
    hipLaunchKernelGGL(svps::slot_attn_stats, dim3(p.chunks, T), dim3(512), svps::StatsLds::total, stream,
                       static_cast<const __bf16*>(q), static_cast<const __bf16*>(k), stats, L, HW, p.tiles_per_chunk);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = launch_partial_ws<SPLIT, 0, true>(q, k, v, partial, T, 128, HW, p, stream, L, 0, stats);
    if (e != hipSuccess) return e;
    return launch_partial_ws<SPLIT, 0, true>(q, k, v, partial, T, L - 128, HW, p, stream, L, 128, stats);
}

size_t stats_bytes(int T, int L, int HW) { return L > 128 ? (size_t)T * HW * sizeof(float2) : 0; }

}  // namespace

extern "C" size_t svps_slot_attn_workspace_bytes(int T, int L, int HW, int chunks) {
    if (T <= 0 || L <= 0 || HW <= 0) return 0;
    const AttnPlan p = plan_attn(T, L, HW, chunks);
    return (size_t)T * p.chunks * L * svps::kD * sizeof(float) + stats_bytes(T, L, HW);
}

extern "C" int svps_slot_attn_plan(int T, int L, int HW, int chunks, int* out_chunks,
                                   int* out_tiles_per_chunk, int* out_tile_px) {
    if (T <= 0 || L <= 0 || HW <= 0) return SVPS_ERR_BAD_ARG;
    const AttnPlan p = plan_attn(T, L, HW, chunks);
    if (out_chunks) *out_chunks = p.chunks;
    if (out_tiles_per_chunk) *out_tiles_per_chunk = p.tiles_per_chunk;
    if (out_tile_px) *out_tile_px = p.tile_px;
    return 0;
}

extern "C" int svps_slot_attn_fwd(const void* q, const void* k, const void* v, const float* ln_w,
                                  const float* ln_b, float ln_eps, void* workspace,
                                  size_t workspace_bytes, float* out, float* out_pre_ln, int T, int L,
                                  int HW, int D, int flags, int chunks, void* stream_) {
    if (!q || !k || !v || !ln_w || !ln_b || !workspace || !out) return SVPS_ERR_BAD_ARG;
    if (D != svps::kD || T <= 0 || L <= 0 || L > 256 || HW <= 0) return SVPS_ERR_BAD_SHAPE;
    if ((size_t)HW > svps::kMaxFramePixels) return SVPS_ERR_BAD_SHAPE;       // 32-bit buffer offsets inside a frame
    const AttnPlan p = plan_attn(T, L, HW, chunks);
    const size_t partial_bytes = (size_t)T * p.chunks * L * svps::kD * sizeof(float);
    if (workspace_bytes < partial_bytes + stats_bytes(T, L, HW)) return SVPS_ERR_WORKSPACE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    float* partial = static_cast<float*>(workspace);
    const bool split = flags & SVPS_FLAG_SPLIT_P;
    hipError_t e;
    svps_prof_mark(SVPS_KERNEL_SLOT_ATTN, 0, stream);
    static const int ablate = [] { const char* a = getenv("SVPS_ABLATE"); return a ? atoi(a) : 0; }();
    if (L <= 128 && ablate) {
        switch (ablate) {
            case 1: e = launch_partial_ws<true, 1>(q, k, v, partial, T, L, HW, p, stream); break;
            case 2: e = launch_partial_ws<true, 2>(q, k, v, partial, T, L, HW, p, stream); break;
            case 4: e = launch_partial_ws<true, 4>(q, k, v, partial, T, L, HW, p, stream); break;
            default: e = launch_partial_ws<true, 5>(q, k, v, partial, T, L, HW, p, stream); break;
        }
    } else if (L <= 128)
        e = split ? launch_partial_ws<true>(q, k, v, partial, T, L, HW, p, stream)
                  : launch_partial_ws<false>(q, k, v, partial, T, L, HW, p, stream);
    else if (getenv("SVPS_K1_LEGACY"))
        e = split ? launch_partial<8, 3, true>(q, k, v, partial, T, L, HW, p, stream)
                  : launch_partial<8, 3, false>(q, k, v, partial, T, L, HW, p, stream);
    else {
        float2* st = reinterpret_cast<float2*>(static_cast<char*>(workspace) + partial_bytes);
        e = split ? launch_two_pass<true>(q, k, v, partial, st, T, L, HW, p, stream)
                  : launch_two_pass<false>(q, k, v, partial, st, T, L, HW, p, stream);
    }
    svps_prof_mark(SVPS_KERNEL_SLOT_ATTN, 1, stream);
    if (e != hipSuccess) return (int)e;
    svps_prof_mark(SVPS_KERNEL_SLOT_ATTN_FINISH, 0, stream);
    hipLaunchKernelGGL(svps::slot_attn_finish, dim3(L, T), dim3(256), 0, stream, partial, ln_w, ln_b, ln_eps,
                       out, out_pre_ln, L, p.chunks);
    svps_prof_mark(SVPS_KERNEL_SLOT_ATTN_FINISH, 1, stream);
    return (int)hipGetLastError();
}
