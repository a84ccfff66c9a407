// K2 - slot -> mask dot-product decode for gfx950.
//
// Replaces VPS_Temporal_Slots.generate_final_outputs (main branch) of the reference,
// mmdet/models/detectors/vps_temporal_slots.py:144-160:
//
//     g[p, :]  = feat_bn(f)[p, :]            = scale (.) f[p, :] + shift        (eval BatchNorm2d)
//     gh[p, :] = g[p, :] / max(||g[p, :]||_2, 1e-12)                            (F.normalize)
//     m[l, p]  = sum_c gh[p, c] * e[l, c]                                       (einsum)
//     m[l, p]  = fg_scale * m[l, p] + fg_shift                                  (fg_bn, scalar)
//
// Folded so that the MFMA runs on the raw bf16 feature tile exactly as it arrives from HBM:
//     m[l, p] = ( sum_c (e[l,c] * scale[c]) * f[p,c]  +  sum_c e[l,c] * shift[c] ) / max(||g_p||, 1e-12)
// The slot operand e (.) scale is carried as bf16 hi + lo (two MFMAs), so the only rounding left is
// fp32 accumulation; the feature map itself is the bf16 tensor the head stores.
//
// HBM-bound: reads T*HW*512 B, writes T*L*HW*4 B. Layout and staging are those of K1 (32-pixel
// tiles by LDS-DMA into a 4-stage ring, wave w owns slots [32w, 32w+32), pixel = lane), so the
// logits leave the accumulators as 128-B pixel-contiguous row segments of the [T, L, HW] output.
#include <cstdlib>
#include <stdlib.h>

#include <type_traits>
#include "common.h"
#include "../../include/slotvps_hip.h"

namespace svps {

template <int NW, int NST, bool HL = false>
struct DecLds {
    static constexpr int ring = 0;                         // NST feature tiles
    static constexpr int lo_ring = NST * kTileBytes;       // HL: NST tiles of the map's lo plane
    static constexpr int affine = (HL ? 2 : 1) * NST * kTileBytes;   // scale[256], shift[256] fp32
    static constexpr int norm = affine + 2 * kD * 4;       // inv-norm per tile pixel [32]
    static constexpr int cshift = norm + kTilePx * 4;      // per-slot constant [NW * 32]
    static constexpr int amax = cshift + NW * 32 * 4;      // per-wave argmax candidates [NW][32] float2
    static constexpr int total = amax + NW * kTilePx * 8;
};

// HL (round 4, the reference-precision mode: svps_mask_decode_hl_fwd): the map arrives as fp16 hi + lo planes (f = hi + lo to 22 bits,
// svps_level_fuse_hl_fwd); both planes of a tile are staged, the norm runs on hi + lo in fp32 and the logits take three MFMAs per
// k-step: e_hi f_hi + e_lo f_hi + e_hi f_lo (e . scale as fp16 hi + lo).
template <int NW, int NST, bool ARGMAX, typename OutT, typename MT = __bf16, bool HL = false>
__global__ __launch_bounds__(NW * 64, (HL && NW == 4) ? 2 : 1) void mask_decode_kernel(
    const MT* __restrict__ feat,     // [T, HW, 256]
    const float* __restrict__ embed,     // [T, L, 256]
    const float* __restrict__ bn_scale,  // [256]
    const float* __restrict__ bn_shift,  // [256]
    float fg_scale, float fg_shift,
    OutT* __restrict__ out,              // [T, L, HW]
    uint8_t* __restrict__ slot_argmax,   // [T, HW] or null
    int L, int HW, int tiles_per_chunk,
    const MT* __restrict__ feat_lo = nullptr) {   // HL: the lo plane [T, HW, 256]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef MT mx8 __attribute__((ext_vector_type(8)));         // MT: element type of the fused map, bf16 or fp16 (common.h)
    using Lds = DecLds<NW, NST, HL>;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int t = blockIdx.y, c = blockIdx.x;

    const int px_begin = c * tiles_per_chunk * kTilePx;
    int px_end = px_begin + tiles_per_chunk * kTilePx;
    px_end = px_end < HW ? px_end : HW;
    const int nt = (px_end - px_begin + kTilePx - 1) / kTilePx;

    float* aff = reinterpret_cast<float*>(smem + Lds::affine);
    float* inv_norm = reinterpret_cast<float*>(smem + Lds::norm);
    float* cs = reinterpret_cast<float*>(smem + Lds::cshift);
    float2* am = reinterpret_cast<float2*>(smem + Lds::amax);

    for (int i = tid; i < kD; i += NW * 64) {
        aff[i] = bn_scale[i];
        aff[kD + i] = bn_shift[i];
    }
    __syncthreads();

    // ---- slot operand (e * scale) as bf16 hi/lo A fragments; per-slot constant e . shift ------
    mx8 eh[16], el[16];
    {
        const int slot = 32 * w + r;
        const float* erow = embed + ((size_t)t * L + (slot < L ? slot : 0)) * kD + 8 * h;
        float dot = 0.f;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(erow + 16 * ks);
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(erow + 16 * ks + 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ch = 16 * ks + 8 * h + j;
                float x = j < 4 ? x0[j] : x1[j - 4];
                if (slot >= L) x = 0.f;
                dot += x * aff[kD + ch];
                float xs = x * aff[ch];
                // xs is ONE fp32 value for both halves: left to itself hipcc rounds hi from the fp32 product (v_cvt_pk_f16_f32) and lo from
                // the exact product (v_fma_mixlo_f16 x, a, -hi') with its own hi' - at a tie the two differ by an fp16 ulp (seen: 2.5e-5 on a logit)
                asm volatile("" : "+v"(xs));
                const MT hi = (MT)xs;
                eh[ks][j] = hi;
                el[ks][j] = (MT)(xs - (float)hi);
            }
        }
        dot = wave_half_xor_sum(dot);
        if (h == 0) cs[32 * w + r] = dot;
    }
    wait_vm<0>();
    __syncthreads();
    float csr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) csr[i] = cs[32 * w + acc_row(i, h)];

    const char* fb = reinterpret_cast<const char*>(feat) + (size_t)t * HW * kRowBytes;
    const char* fbl = HL ? reinterpret_cast<const char*>(feat_lo) + (size_t)t * HW * kRowBytes : nullptr;
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nt) {
            dma_tile<NW>(fb, px_begin + s * kTilePx, HW - 1, smem + Lds::ring + s * kTileBytes, w, lane);
            if constexpr (HL) dma_tile<NW>(fbl, px_begin + s * kTilePx, HW - 1, smem + Lds::lo_ring + s * kTileBytes, w, lane);
        }

    constexpr int PIECES = (HL ? 2 : 1) * (kTilePx / 2 / NW);   // DMA instructions per wave and tile
    // thread -> (pixel, channel octet set) for the norm pass: 8 * NW / 4 threads per pixel
    constexpr int TPP = NW * 64 / kTilePx;       // threads per pixel (8 or 16)
    constexpr int CPT = 32 / TPP;                // 16-B chunks per thread (4 or 2)
    const int npx = tid / TPP, nsub = tid % TPP;

    for (int it = 0; it < nt; ++it) {
        if constexpr (NST == 4) {
            if (it + 2 < nt) wait_vm<2 * PIECES>();
            else if (it + 1 < nt) wait_vm<PIECES>();
            else wait_vm<0>();
        } else {
            wait_vm<0>();
        }
        wg_barrier();
        if (it + NST - 1 < nt) {
            dma_tile<NW>(fb, px_begin + (it + NST - 1) * kTilePx, HW - 1,
                         smem + Lds::ring + ((it + NST - 1) % NST) * kTileBytes, w, lane);
            if constexpr (HL)
                dma_tile<NW>(fbl, px_begin + (it + NST - 1) * kTilePx, HW - 1,
                             smem + Lds::lo_ring + ((it + NST - 1) % NST) * kTileBytes, w, lane);
        }
        const char* ft = smem + Lds::ring + (it % NST) * kTileBytes;
        const char* ftl = smem + Lds::lo_ring + (it % NST) * kTileBytes;

        // -- ||scale * f + shift||^2 per pixel -------------------------------------------------
        {
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                const int chunk = nsub + TPP * i;
                const mx8 x = *reinterpret_cast<const mx8*>(ft + npx * kRowBytes + ((chunk ^ swz(npx)) * 16));
                mx8 xl;
                if constexpr (HL) xl = *reinterpret_cast<const mx8*>(ftl + npx * kRowBytes + ((chunk ^ swz(npx)) * 16));
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xv = HL ? (float)x[j] + (float)xl[j] : (float)x[j];
                    const float g = xv * aff[8 * chunk + j] + aff[kD + 8 * chunk + j];
                    ss += g * g;
                }
            }
#pragma unroll
            for (int m = 1; m < TPP; m <<= 1) ss += __shfl_xor(ss, m);
            if (nsub == 0) inv_norm[npx] = 1.f / fmaxf(sqrtf(ss), 1e-12f);
        }

        // -- (e * scale) . f for 32 slots x 32 pixels --------------------------------------------
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const mx8 ff = read_row_frag_as<mx8>(ft, ks, r, h);
            s = mfma16(el[ks], ff, s);
            if constexpr (HL) s = mfma16(eh[ks], read_row_frag_as<mx8>(ftl, ks, r, h), s);
            s = mfma16(eh[ks], ff, s);
        }
        wg_barrier();  // inv_norm of this tile visible

        const int px = px_begin + it * kTilePx + r;
        const float inr = inv_norm[r];
        const bool pix_ok = px < px_end;
        float best = -INFINITY;
        int best_slot = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int slot = 32 * w + acc_row(i, h);
            const float m = (s[i] + csr[i]) * inr * fg_scale + fg_shift;
            if (slot < L) {
                if (pix_ok) out[((size_t)t * L + slot) * HW + px] = (OutT)m;
                if constexpr (ARGMAX) {
                    if (m > best) { best = m; best_slot = slot; }  // slots ascend with i within a lane
                }
            }
        }
        if constexpr (ARGMAX) {
            // lanes r and r+32 hold interleaved slot groups of the same pixel
            const float ob = __shfl_xor(best, 32);
            const int os = __shfl_xor(best_slot, 32);
            if (ob > best || (ob == best && os < best_slot)) { best = ob; best_slot = os; }
            if (h == 0) am[w * kTilePx + r] = make_float2(best, __int_as_float(best_slot));
            wg_barrier();
            if (w == 0 && h == 0 && pix_ok) {
                float b = -INFINITY;
                int bs = 0x7fffffff;
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) {
                    const float2 cnd = am[ww * kTilePx + r];
                    const int sl = __float_as_int(cnd.y);
                    if (cnd.x > b || (cnd.x == b && sl < bs)) { b = cnd.x; bs = sl; }
                }
                slot_argmax[(size_t)t * HW + px] = (uint8_t)bs;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Fast path (L <= 128, HW % 4 == 0, fp32 out): same math and tile machinery, three changes that matter on gfx950
// (every 1-KiB vector-memory instruction stalls its wave for ~75-200 cycles; hipcc drains the builtin DMA form):
//   * LDS-DMA in inline asm (invisible to hipcc's counters), one M0 write per tile and wave, counted vmcnt;
//   * the logits leave through a wave-local LDS transpose: 16 ds_write_b32 per lane, then four 16-byte-per-lane
//     buffer stores per wave and tile (8 lanes = one 128-byte row segment of a slot) instead of 16 dword stores;
//     invalid lanes (slot >= L, pixels past the chunk) get an out-of-range offset and are dropped by the hardware
//     range check, so every wave issues the same number of stores and the vmcnt arithmetic stays exact;
//   * the cross-wave part of the fused argmax of tile j runs at the top of tile j + 1 (double-buffered candidates):
//     two workgroup barriers per tile instead of three.
template <int NW, bool LOGITS = true, int NSTG = 0>
struct Dec2Lds {
    static constexpr int kStages = NSTG ? NSTG : (LOGITS ? 3 : 4);   // argmax-only mode has no transpose tiles: one more feature tile in flight
    static constexpr int ring = 0;                                   // kStages feature tiles
    static constexpr int kORow = 144;                                // per wave [32 slots][32 px] fp32, rows padded to 144 B:
    static constexpr int kOWave = 32 * kORow;                        // every epilogue address is lane base + constant
    static constexpr int otile = kStages * kTileBytes;
    static constexpr int affine = otile + (LOGITS ? NW * kOWave : 0);   // scale[256], shift[256]
    static constexpr int norm = affine + 2 * kD * 4;                 // [32]
    static constexpr int cshift = norm + 2 * kTilePx * 4;            // [32 NW]   (norm: [2][32], by tile parity)
    static constexpr int amax = cshift + 32 * NW * 4;                // [2][NW][32] float2
    static constexpr int total = amax + 2 * NW * kTilePx * 8;
};

__device__ __forceinline__ u32x4 make_srd_d(const void* base, uint32_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    u32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
    d[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
    d[2] = __builtin_amdgcn_readfirstlane(bytes);
    d[3] = 0x00020000u;
    return d;
}

__device__ __forceinline__ void dma16_d(u32x4 srd, uint32_t lds_addr, int voff, int soff) {
    uint32_t keep;      // M0 (LDS base of the DMA) is saved and restored inside the statement
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

// four pieces, one M0 write: the instruction offset advances the LDS address and the global address together
__device__ __forceinline__ void dma16x4_d(u32x4 srd, uint32_t lds_addr, int v0, int v1, int v2, int v3, int soff) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        // nt: the map is streamed once (3.5 % on this kernel; nt on the mask stores costs as much)
        "buffer_load_dwordx4 %2, %6, %7 offen nt lds\n\t"
        "buffer_load_dwordx4 %3, %6, %7 offen offset:1024 nt lds\n\t"
        "buffer_load_dwordx4 %4, %6, %7 offen offset:2048 nt lds\n\t"
        "buffer_load_dwordx4 %5, %6, %7 offen offset:3072 nt lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_addr), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(srd), "s"(soff)
        : "memory");
}

// counted like the DMA: asm, so that hipcc neither reorders nor skips it (exact vmcnt arithmetic)
__device__ __forceinline__ void store16_d(u32x4 val, u32x4 srd, int voff) {
    // s_nop: gfx950 reads the data VGPRs of a >64-bit store late; hipcc's hazard pass does not see inside asm
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" : : "v"(val), "v"(voff), "s"(srd) : "memory");
}
__device__ __forceinline__ void store1_d(int val, u32x4 srd, int voff) {
    asm volatile("buffer_store_byte %0, %1, %2, 0 offen" : : "v"(val), "v"(voff), "s"(srd) : "memory");
}

#ifdef SVPS_K2_STAMP
// diagnostic build only (tools/k2_stamps.py): s_memtime stamps of the four waves of one workgroup, iterations 8 .. 15
__device__ unsigned long long k2_stamps[4][8][8];            // [wave][iteration - 8][point]
#define K2_STAMP(pt)                                                                                     \
    do {                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        if (blockIdx.x == 3 && blockIdx.y == 2 && it >= 8 && it < 16 && w < 4 && (threadIdx.x & 63) == 0) \
            k2_stamps[w][it - 8][pt] = __builtin_amdgcn_s_memtime();                                     \
        __builtin_amdgcn_sched_barrier(0);                                                               \
    } while (0)
#else
#define K2_STAMP(pt) do {} while (0)
#endif

// NW waves = 32 NW slots: 4 (L <= 128: two workgroups per CU) or 8 (L <= 256: one workgroup of 512 threads per CU)
// LOGITS = false: argmax-only mode (out == NULL) - the [T, L, HW] logits are neither transposed nor stored: per pixel 512 B in and
// 1 B out instead of 512 + 4 L + 1 (a consumer that only needs the per-pixel slot id, e.g. the clip driver's assignment map)
// ABL (timing-only builds, -DSVPS_K2_ABLATE + tools/ablate_k2.sh; the HL form: tools/kbench_k2hl.py with SVPS_K2_ABLATE): 1 no MFMAs, 2 no
// fragment reads either, 4 no epilogue (the chain is then dead code as well: = 6), 8 no DMA, 64 no logit stores (round 5, HL at T = 40: base 2 120 us,
// 64: 2 119 - the stores are free -, 2: 1 776, 6: 840, 14: 514)
// HL (round 5, the reference-precision mode; NW = 4, MT = fp16): the map arrives as fp16 hi + lo planes. As in retr_attn_kernel<.., HL> a tile
// is SIXTEEN pixels - LDS rows 0 .. 15 their hi rows (staged by waves 0, 1 from `feat`), rows 16 .. 31 their lo rows (waves 2, 3 from
// `feat_lo`) - so ring, swizzle, fragment addresses, barrier schedule and vmcnt arithmetic are those of the 32-pixel form. The chain
// (e as fp16 hi + lo) yields [e.f_hi | e.f_lo] in the two column halves of the accumulator, folded with one v_permlane16_swap + add per
// register; the norm runs on hi + lo in fp32 (16 threads per pixel); the logits leave as two 16-byte stores per lane and tile (4 lanes =
// one 64-byte row segment of a slot; consecutive tiles fill the other half of the 128-byte line).
template <bool ARGMAX, int NW = 4, bool LOGITS = true, int ABL = 0, int NSTG = 0, typename MT = __bf16, bool HL = false>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void mask_decode_kernel_v2(
    const MT* __restrict__ feat, const float* __restrict__ embed, const float* __restrict__ bn_scale,
    const float* __restrict__ bn_shift, float fg_scale, float fg_shift, float* __restrict__ out,
    uint8_t* __restrict__ slot_argmax, int L, int HW, int tiles_per_chunk, const MT* __restrict__ feat_lo = nullptr) {
    static_assert(!HL || (NW == 4 && !(ABL & 32)), "the hi + lo form: four waves, the skewed loop");
    constexpr int TPX = HL ? 16 : kTilePx;                        // pixels per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef MT mx8 __attribute__((ext_vector_type(8)));         // MT: element type of the fused map, bf16 or fp16 (common.h)
    using Lds = Dec2Lds<NW, LOGITS, NSTG>;
    constexpr int NST = Lds::kStages;
    constexpr int kAhead = NST - 1;                              // tiles requested ahead of the one in work
    constexpr int NT = 64 * NW;                                  // threads
    constexpr int PC = 16 / NW;                                  // 1-KiB DMA pieces per wave and tile
    constexpr int PCW = (ABL & 8) ? 0 : PC;                      // ... that are in flight (vmcnt arithmetic)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r_ = lane & 31, h_ = lane >> 5;
    int r = r_, h = h_;
    const int t = blockIdx.y, c = blockIdx.x;

    const int px_begin = c * tiles_per_chunk * TPX;
    int px_end = px_begin + tiles_per_chunk * TPX;
    px_end = px_end < HW ? px_end : HW;
    const int nt = (px_end - px_begin + TPX - 1) / TPX;

    float* aff = reinterpret_cast<float*>(smem + Lds::affine);
    float* inv_norm = reinterpret_cast<float*>(smem + Lds::norm);
    float* cs = reinterpret_cast<float*>(smem + Lds::cshift);
    float2* am = reinterpret_cast<float2*>(smem + Lds::amax);

    for (int i = tid; i < kD; i += NT) {
        aff[i] = bn_scale[i];
        aff[kD + i] = bn_shift[i];
    }
    __syncthreads();

    mx8 eh[16], el[16];
    {
        const int slot = 32 * w + r;
        const float* erow = embed + ((size_t)t * L + (slot < L ? slot : 0)) * kD + 8 * h;
        float dot = 0.f;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(erow + 16 * ks);
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(erow + 16 * ks + 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ch = 16 * ks + 8 * h + j;
                float x = j < 4 ? x0[j] : x1[j - 4];
                if (slot >= L) x = 0.f;
                dot += x * aff[kD + ch];
                float xs = x * aff[ch];
                // xs is ONE fp32 value for both halves: left to itself hipcc rounds hi from the fp32 product (v_cvt_pk_f16_f32) and lo from
                // the exact product (v_fma_mixlo_f16 x, a, -hi') with its own hi' - at a tie the two differ by an fp16 ulp (seen: 2.5e-5 on a logit)
                asm volatile("" : "+v"(xs));
                const MT hi = (MT)xs;
                eh[ks][j] = hi;
                el[ks][j] = (MT)(xs - (float)hi);
            }
            // four k-steps of loads in flight, not sixteen: with all 128 loaded floats live next to the 128 operand registers hipcc
            // spills lane constants, reloads one inside the tile loop and guards it with s_waitcnt vmcnt(0) - which drains the
            // LDS-DMA ring (invisible to the compiler) once per tile
            if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        dot = wave_half_xor_sum(dot);
        if (h == 0) cs[32 * w + r] = dot;
    }
    wait_vm<0>();
    __syncthreads();
    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>((SVPS_LDS const void*)smem);
    const u32x4 frs = make_srd_d((HL && w >= 2 ? feat_lo : feat) + (size_t)t * HW * kD, (uint32_t)HW * kRowBytes);
    const u32x4 ors = make_srd_d(LOGITS ? out + (size_t)t * L * HW : nullptr, LOGITS ? (uint32_t)L * (uint32_t)HW * 4u : 0u);
    constexpr int kMS = LOGITS ? (HL ? 2 : 4) : 0;               // mask stores per wave and tile
    const u32x4 ars = make_srd_d(ARGMAX ? slot_argmax + (size_t)t * HW : nullptr, ARGMAX ? (uint32_t)HW : 0u);
    auto stage = [&](int tile) {          // exactly PC DMA instructions per wave, or none
        if (tile >= nt || (ABL & 8)) return;
        int voff[PC];                     // recomputed per tile (a handful of VALU ops) rather than held in VGPRs
        {
            int rr = r_, hh = h_;
            asm volatile("" : "+v"(rr), "+v"(hh));
#pragma unroll
            for (int i = 0; i < PC; ++i) {
                const int row = 2 * PC * w + 2 * i + hh;
                voff[i] = (HL ? (row & 15) : row) * kRowBytes + ((rr ^ swz(row)) * 16);
            }
        }
        const uint32_t st = __builtin_amdgcn_readfirstlane(lds0 + Lds::ring + (tile % NST) * kTileBytes + w * PC * 1024);
        const int px0 = px_begin + tile * TPX;
        const int soff = __builtin_amdgcn_readfirstlane(px0 * kRowBytes);
        if (px0 + TPX <= HW) {
            if constexpr (PC == 4) {
                dma16x4_d(frs, st, voff[0], voff[1] - 1024, voff[2] - 2048, voff[3] - 3072, soff);
            } else {
#pragma unroll
                for (int i = 0; i < PC; ++i) dma16_d(frs, st + i * 1024, voff[i], soff);
            }
        } else {
#pragma unroll
            for (int i = 0; i < PC; ++i) {
                const int row = 2 * PC * w + 2 * i + h;
                const int prow = HL ? (row & 15) : row;
                const int src = px0 + prow < HW ? prow : HW - 1 - px0;
                dma16_d(frs, st + i * 1024, src * kRowBytes + (((lane & 31) ^ swz(row)) * 16), soff);
            }
        }
    };
    const int amw = (ARGMAX && w == 0) ? 1 : 0;          // wave 0 also stores the argmax bytes: one more store per tile
    auto finish_argmax = [&](int tile) {                 // cross-wave part of the argmax of `tile`, wave 0
        if (h == 0) {
            float b = -INFINITY;
            int bs = 0x7fffffff;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) {
                const float2 cnd = am[(tile & 1) * NW * kTilePx + ww * kTilePx + r];
                const int sl = __float_as_int(cnd.y);
                if (cnd.x > b || (cnd.x == b && sl < bs)) { b = cnd.x; bs = sl; }
            }
            const int px = px_begin + tile * TPX + r;
            store1_d(bs, ars, (px < px_end && r < TPX) ? px : 0x7ffffff0);
        } else {
            store1_d(0, ars, 0x7ffffff0);                // same instruction for every lane; dropped by the range check
        }
    };

#pragma unroll
    for (int i = 0; i < kAhead; ++i) stage(i);

    char* ot = smem + Lds::otile + w * Lds::kOWave;

    if constexpr (ABL & 32) {   // the un-skewed loop (round 2 / start of round 3), kept in ablation builds for same-box A/B
        for (int it = 0; it < nt; ++it) {
            // tile `it` landed (this wave's pieces). Younger than its DMA, in issue order (kAhead = 2): argmax(it-3), stores(it-2),
            // DMA(it+1), argmax(it-2), stores(it-1); in general the stores of the last kAhead iterations and the DMA of kAhead - 1 tiles.
            {
                int younger = 0;
    #pragma unroll
                for (int j = 1; j <= kAhead; ++j)
                    younger += kMS * (it >= j) + amw * (it >= j + 1) + (j < kAhead ? PCW * (it + j < nt) : 0);
                wait_vm_dyn(younger);
            }
            wg_barrier();
            stage(it + kAhead);
            int tid_o = tid, lane_o = lane;
            r = r_; h = h_;
            asm volatile("" : "+v"(r), "+v"(h), "+v"(tid_o), "+v"(lane_o));   // opaque per tile: no loop-invariant address tables in VGPRs
            if (ARGMAX && w == 0 && it >= 1) finish_argmax(it - 1);
            const char* ft = smem + Lds::ring + (it % NST) * kTileBytes;

            // Argmax-only mode: m = (s + c) * inr * fg_scale + fg_shift is a monotone function of s + c for a pixel (inr > 0), so
            // the per-pixel norm is not needed to order the slots - the argmax is taken over sgn(fg_scale) * (s + c). It equals the
            // full mode's argmax except where two slots' logits round to the SAME fp32 value (the full mode then reports the lower
            // slot, this mode the larger s + c): pixels without a decision at fp32 resolution.
            if constexpr (LOGITS) {   // ||scale * f + shift||^2 per pixel: 2 NW threads per pixel, 16 / NW chunks each
                constexpr int TPP = 2 * NW;
                const int npx = tid_o / TPP, nsub = tid_o % TPP;
                float ss = 0.f;
    #pragma unroll
                for (int i = 0; i < 32 / TPP; ++i) {
                    const int chunk = nsub + TPP * i;
                    const mx8 x = *reinterpret_cast<const mx8*>(ft + npx * kRowBytes + ((chunk ^ swz(npx)) * 16));
    #pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float g = (float)x[j] * aff[8 * chunk + j] + aff[kD + 8 * chunk + j];
                        ss += g * g;
                    }
                    __builtin_amdgcn_sched_barrier(0);       // one chunk at a time: the affine rows are not worth 64 VGPRs
                }
                ss += __shfl_xor(ss, 1);
                ss += __shfl_xor(ss, 2);
                ss += __shfl_xor(ss, 4);
                if constexpr (TPP == 16) ss += __shfl_xor(ss, 8);
                if (nsub == 0) inv_norm[npx] = 1.f / fmaxf(sqrtf(ss), 1e-12f);
            }

            f32x16 s;
    #pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = 0.f;
            {
    #pragma unroll
                for (int grp = 0; grp < ((ABL & 2) ? 0 : 4); ++grp) {   // four operand fragments in flight (register budget: 128 hold e)
                    mx8 ff[4];
    #pragma unroll
                    for (int u = 0; u < 4; ++u) ff[u] = read_row_frag_as<mx8>(ft, 4 * grp + u, r, h);
                    __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if constexpr (ABL & 1) {
                            asm volatile("" : : "v"(ff[u]), "v"(el[4 * grp + u]), "v"(eh[4 * grp + u]));
                        } else {
                            s = mfma16(el[4 * grp + u], ff[u], s);
                            s = mfma16(eh[4 * grp + u], ff[u], s);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            float inr = 1.f;
            if constexpr (LOGITS) {
                wg_barrier();                                // inv_norm of this tile visible
                inr = inv_norm[r];
            }
            const float ksgn = fg_scale > 0.f ? 1.f : (fg_scale < 0.f ? -1.f : 0.f);
            float best = -INFINITY;
            int best_slot = 0x7fffffff;
    #pragma unroll
            for (int i = 0; i < ((ABL & 4) ? 1 : 16); ++i) {
                const int sl = acc_row(i, h);                // slot inside this wave's block of 32
                f32x4 c4;
                if ((i & 3) == 0) c4 = *reinterpret_cast<const f32x4*>(cs + 32 * w + sl);   // e . shift of slots sl .. sl + 3
                const float m = LOGITS ? (s[i] + c4[i & 3]) * inr * fg_scale + fg_shift : (s[i] + c4[i & 3]) * ksgn;
                if constexpr (LOGITS) *reinterpret_cast<float*>(ot + sl * Lds::kORow + r * 4) = m;
                if constexpr (ARGMAX) {
                    if (32 * w + sl < L && m > best) { best = m; best_slot = 32 * w + sl; }   // slots ascend with i within a lane
                }
            }
            if constexpr (ARGMAX) {
                const float ob = __shfl_xor(best, 32);
                const int os = __shfl_xor(best_slot, 32);
                if (ob > best || (ob == best && os < best_slot)) { best = ob; best_slot = os; }
                if (h == 0) am[(it & 1) * NW * kTilePx + w * kTilePx + r] = make_float2(best, __int_as_float(best_slot));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-local transpose: own writes done, no barrier needed
            const int px0 = px_begin + it * kTilePx;
    #pragma unroll
            for (int u = 0; u < kMS; ++u) {
                const int sl = 8 * u + (lane_o >> 3), cc = lane_o & 7;
                const u32x4 val = *reinterpret_cast<const u32x4*>(ot + sl * Lds::kORow + cc * 16);
                const int slot = 32 * w + sl, px = px0 + 4 * cc;
                const bool ok = slot < L && px < px_end && !(ABL & 64);       // ABL 64 (timing only): no logit stores
                store16_d(val, ors, ok ? (slot * HW + px) * 4 : 0x7ffffff0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (ARGMAX) {
            wg_barrier();
            if (w == 0) finish_argmax(nt - 1);
        }
        return;
    }

    // Skewed loop: the epilogue of tile it - 1 (constants, scale, argmax candidates, transpose writes) rides in the shadow of the
    // MFMA chain of tile it - one accumulator element per MFMA pair - and the operand fragments of group g + 1 are requested before
    // the MFMAs of group g. One barrier per tile in either mode (the per-pixel norms of tile it are published by the barrier of
    // iteration it + 1, double-buffered by parity). Iteration 0 has no epilogue, iteration nt no chain.
    f32x16 sp;                                            // accumulators of the previous tile
#pragma unroll
    for (int i = 0; i < 16; ++i) sp[i] = 0.f;
    const float ksgn = fg_scale > 0.f ? 1.f : (fg_scale < 0.f ? -1.f : 0.f);
    constexpr int kAS = (ARGMAX && !(ABL & 4)) ? 1 : 0;                    // argmax stores per wave and iteration (only wave 0's lanes are in range)
    auto body = [&](const int it, auto has_chain, auto has_epi) {
        constexpr bool CH = decltype(has_chain)::value, EP = decltype(has_epi)::value && !(ABL & 4);
        // Per iteration it >= 1, in issue order: DMA(it + kAhead), argmax store of tile it - 2, mask stores of tile it - 1.
        // Younger than DMA(it): the stores of the last kAhead iterations and the DMA of kAhead - 1 tiles.
        K2_STAMP(0);
        if constexpr (CH) {
            constexpr int kSteady = kAhead * (kMS + kAS) + (kAhead - 1) * PCW;
            if (it > kAhead && it + kAhead <= nt) {        // steady state: a constant, no branch tree
                wait_vm<kSteady>();
            } else {
                int younger = 0;
#pragma unroll
                for (int m = 1; m <= kAhead; ++m)
                    younger += (kMS + kAS) * (it - m >= 1) + (m < kAhead ? PCW * (it - m + kAhead < nt) : 0);
                wait_vm_dyn(younger);
            }
        }
        K2_STAMP(1);
        wg_barrier();
        K2_STAMP(2);
        if constexpr (CH) stage(it + kAhead);
        K2_STAMP(3);
        int tid_o = tid, lane_o = lane;
        r = r_; h = h_;
        asm volatile("" : "+v"(r), "+v"(h), "+v"(tid_o), "+v"(lane_o));   // opaque per tile: no loop-invariant address tables in VGPRs
        const char* ft = smem + Lds::ring + (it % NST) * kTileBytes;

        // Argmax-only mode: m = (s + c) * inr * fg_scale + fg_shift is a monotone function of s + c for a pixel (inr > 0), so
        // the per-pixel norm is not needed to order the slots - the argmax is taken over sgn(fg_scale) * (s + c). It equals the
        // full mode's argmax except where two slots' logits round to the SAME fp32 value (the full mode then reports the lower
        // slot, this mode the larger s + c): pixels without a decision at fp32 resolution.
        if constexpr (LOGITS && CH) {   // ||scale * f + shift||^2 per pixel: 2 NW threads per pixel, 16 / NW chunks each (HL: 16 threads, 2 chunks)
            constexpr int TPP = HL ? 16 : 2 * NW;
            const int npx = tid_o / TPP, nsub = tid_o % TPP;
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < 32 / TPP; ++i) {
                const int chunk = nsub + TPP * i;
                const mx8 x = *reinterpret_cast<const mx8*>(ft + npx * kRowBytes + ((chunk ^ swz(npx)) * 16));
                mx8 xl = x;
                if constexpr (HL) xl = *reinterpret_cast<const mx8*>(ft + (npx + 16) * kRowBytes + ((chunk ^ swz(npx + 16)) * 16));
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xv = HL ? (float)x[j] + (float)xl[j] : (float)x[j];
                    const float g = xv * aff[8 * chunk + j] + aff[kD + 8 * chunk + j];
                    ss += g * g;
                }
                __builtin_amdgcn_sched_barrier(0);       // one chunk at a time: the affine rows are not worth 64 VGPRs
            }
            ss += __shfl_xor(ss, 1);
            ss += __shfl_xor(ss, 2);
            ss += __shfl_xor(ss, 4);
            if constexpr (TPP == 16) ss += __shfl_xor(ss, 8);
            if (nsub == 0) inv_norm[(it & 1) * kTilePx + npx] = 1.f / fmaxf(sqrtf(ss), 1e-12f);
        }

        constexpr bool FR = CH && !(ABL & 2);            // fragments are read
        mx8 ff[2][4];
        if constexpr (FR) {
#pragma unroll
            for (int u = 0; u < 4; ++u) ff[0][u] = read_row_frag_as<mx8>(ft, u, r, h);
        }
        // Cross-wave part of the argmax of tile it - 2 (its candidates were written in iteration it - 1): every wave reads them and
        // runs the comparison in the shadow of its first MFMAs - nobody is late at the next barrier; only wave 0's offsets are in
        // range, the other waves' stores (and iteration 1's) are dropped by the range check.
        float2 cnd[NW];
        if constexpr (ARGMAX && EP) {
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) cnd[ww] = am[(it & 1) * NW * kTilePx + ww * kTilePx + r];
        }
        float inr = 1.f;
        f32x4 c4[2];                                     // e . shift of slots sl .. sl + 3, one group ahead
        // HL: after the fold by register exchange (end of the previous iteration) lane r < 16 holds the sums of accumulator registers 0 .. 7
        // and lane r + 16 those of registers 8 .. 15 of the SAME pixel: eight epilogue elements per lane (slot groups 2 hi16, 2 hi16 + 1)
        const int hi16 = HL ? ((r >> 4) & 1) : 0;
        constexpr int kEG = HL ? 2 : 4;                  // epilogue groups of four elements per lane
        if constexpr (EP) {
            if constexpr (LOGITS) inr = inv_norm[((it - 1) & 1) * kTilePx + (HL ? (r & 15) : r)];
            c4[0] = *reinterpret_cast<const f32x4*>(cs + 32 * w + acc_row(0, h) + 16 * hi16);
        }
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
        float best = -INFINITY;
        int best_slot = 0x7fffffff;
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            if constexpr (FR) {
                if (grp < 3) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) ff[(grp + 1) & 1][u] = read_row_frag_as<mx8>(ft, 4 * (grp + 1) + u, r, h);
                }
            }
            if constexpr (EP) {
                if (grp + 1 < kEG) c4[(grp + 1) & 1] = *reinterpret_cast<const f32x4*>(cs + 32 * w + acc_row(4 * (grp + 1), h) + 16 * hi16);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if constexpr (FR) {
                    if constexpr (ABL & 1) {
                        asm volatile("" : : "v"(ff[grp & 1][u]), "v"(el[4 * grp + u]), "v"(eh[4 * grp + u]));
                    } else {
                        s = mfma16(el[4 * grp + u], ff[grp & 1][u], s);
                        s = mfma16(eh[4 * grp + u], ff[grp & 1][u], s);
                    }
                }
                if constexpr (ARGMAX && EP) {
                    if (grp == 0 && u == 1) {
                        float b = -INFINITY;
                        int bs = 0x7fffffff;
#pragma unroll
                        for (int ww = 0; ww < NW; ++ww) {
                            const int sl = __float_as_int(cnd[ww].y);
                            if (cnd[ww].x > b || (cnd[ww].x == b && sl < bs)) { b = cnd[ww].x; bs = sl; }
                        }
                        const int px = px_begin + (it - 2) * TPX + r;
                        const bool ok = w == 0 && h == 0 && it >= 2 && px < px_end && r < TPX;
                        store1_d(bs, ars, ok ? px : 0x7ffffff0);
                    }
                }
                if constexpr (EP) {
                    if (grp < kEG) {
                        const int i = 4 * grp + u;
                        const int sl = acc_row(i, h) + 16 * hi16;   // slot inside this wave's block of 32
                        const float m = LOGITS ? (sp[i] + c4[grp & 1][u]) * inr * fg_scale + fg_shift : (sp[i] + c4[grp & 1][u]) * ksgn;
                        if constexpr (LOGITS) *reinterpret_cast<float*>(ot + sl * Lds::kORow + (HL ? (r & 15) : r) * 4) = m;
                        if constexpr (ARGMAX) {
                            if (32 * w + sl < L && m > best) { best = m; best_slot = 32 * w + sl; }   // slots ascend with i within a lane
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        K2_STAMP(5);
        if constexpr (EP) {
            if constexpr (ARGMAX) {
                if constexpr (HL) {                      // the two lanes of a pixel hold different slots: combine them first (ties: lower slot)
                    const auto qb = __builtin_amdgcn_permlane16_swap(__float_as_uint(best), __float_as_uint(best), false, false);
                    const auto qs = __builtin_amdgcn_permlane16_swap((uint32_t)best_slot, (uint32_t)best_slot, false, false);
                    const float b0 = __uint_as_float(qb[0]), b1 = __uint_as_float(qb[1]);
                    const int s0 = (int)qs[0], s1 = (int)qs[1];
                    const bool take1 = b1 > b0 || (b1 == b0 && s1 < s0);
                    best = take1 ? b1 : b0;
                    best_slot = take1 ? s1 : s0;
                }
                // v_permlane32_swap: [0] = the value of lane r, [1] = of lane r + 32, in both lanes (no address register, no LDS trip)
                const auto pb = __builtin_amdgcn_permlane32_swap(__float_as_uint(best), __float_as_uint(best), false, false);
                const auto ps = __builtin_amdgcn_permlane32_swap((uint32_t)best_slot, (uint32_t)best_slot, false, false);
                best = __uint_as_float(pb[0]);
                best_slot = (int)ps[0];
                const float ob = __uint_as_float(pb[1]);
                const int os = (int)ps[1];
                if (ob > best || (ob == best && os < best_slot)) { best = ob; best_slot = os; }
#ifdef SVPS_K2_STAMP
                asm volatile("" : "+v"(best), "+v"(best_slot));
                K2_STAMP(4);
#endif
                if (h == 0) am[((it - 1) & 1) * NW * kTilePx + w * kTilePx + r] = make_float2(best, __int_as_float(best_slot));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-local transpose: own writes done, no barrier needed
            const int px0 = px_begin + (it - 1) * TPX;
#pragma unroll
            for (int u = 0; u < kMS; ++u) {
                const int sl = HL ? 16 * u + (lane_o >> 2) : 8 * u + (lane_o >> 3), cc = HL ? (lane_o & 3) : (lane_o & 7);
                const u32x4 val = *reinterpret_cast<const u32x4*>(ot + sl * Lds::kORow + cc * 16);
                const int slot = 32 * w + sl, px = px0 + 4 * cc;
                const bool ok = slot < L && px < px_end;
                store16_d(val, ors, ok ? (slot * HW + px) * 4 : 0x7ffffff0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        K2_STAMP(6);
        if constexpr (CH && HL) {
            // columns r < 16: e . f_hi of pixel r; columns r >= 16: e . f_lo of pixel r - 16. Fold by EXCHANGING register halves (as in
            // retr_attn_kernel<.., HL>): v_permlane16_swap(s[i], s[i + 8]) returns [s_i.row0, s_{i+8}.row0, ..] and [s_i.row1, s_{i+8}.row1, ..],
            // whose sum is the full value of register i in lanes r < 16 and of register i + 8 in lanes r >= 16: eight epilogue elements per lane
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(s[i]), __float_as_uint(s[i + 8]), false, false);
                s[i] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            }
        }
        if constexpr (CH) sp = s;
#ifdef SVPS_K2_STAMP
        asm volatile("" : "+v"(sp));
#endif
        K2_STAMP(7);
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    body(0, T_{}, F_{});
#pragma clang loop unroll(disable)
    for (int it = 1; it < nt; ++it) body(it, T_{}, T_{});
    body(nt, F_{}, T_{});
    if constexpr (ARGMAX) {
        wg_barrier();
        if (w == 0) finish_argmax(nt - 1);
    }
}


// ---------------------------------------------------------------------------------------------------------------
// HL32 (round 6): the reference-precision decode on THIRTY-TWO-pixel tiles. mask_decode_kernel_v2<.., HL> borrowed the 16-pixel hi / lo tile
// of the retriever: per 16 pixels it pays a whole tile's fixed cost (barrier, DMA issue, argmax reduction, two half-line stores), a useless
// e_lo . f_lo quarter of its MFMAs and a fold of the two accumulator halves - 17 vector instructions per MFMA, VALU-issue-bound (round 5:
// 0.44 of HBM). Here a tile is 32 pixels with the hi rows and the lo rows in TWO 16-KiB LDS tiles of the usual layout:
//   * three MFMAs per k-step into ONE accumulator (e_lo f_hi + e_hi f_lo + e_hi f_hi): 48 per wave and 32 pixels instead of 64, no fold;
//     lane = pixel, 16 accumulator elements = 16 slots, exactly the layout of the 16-bit fast path;
//   * one barrier, one argmax reduction and four full 128-byte row-segment stores per wave and 32 pixels;
//   * fragment addresses held in 8 registers (chunk ^ swizzle for k-steps 0 .. 7; k-steps 8 .. 15 and the lo tile are immediate offsets);
//   * LDS: 2 stages x 32 KiB + a [16 slots][32 px] transpose tile per wave (the logits leave in two halves) = 78 KiB: two workgroups per CU.
// Same skew as the fast path: the epilogue of tile it - 1 rides in the shadow of the chain of tile it; norms double-buffered by parity.
// NW = 4 waves (up to 128 slots, two workgroups per CU) or 8 (up to 256 slots - VIPER's 200 -, one workgroup per CU: the same eight waves).
#ifndef SVPS_K2_HL32_AHEAD
#define SVPS_K2_HL32_AHEAD 1        // operand fragments requested this many k-steps (of three MFMAs) ahead
#endif
#ifndef SVPS_K2_HL32_W8_STAGES
#define SVPS_K2_HL32_W8_STAGES 2       // (3 measured 1 % slower at the VIPER shape: the eight-wave form is bound by its own work per pixel, not by landing waits)
#endif
template <int NW, int NST = 2>
struct DecHl32LdsT {
    static constexpr int kStages = NST;
    static constexpr int kStageBytes = 2 * kTileBytes;               // hi tile, then lo tile
    static constexpr int ring = 0;
    static constexpr int kORow = 144;                                // [16 slots][32 px] fp32 per wave, rows padded to 144 B
    static constexpr int kOWave = 16 * kORow;
    static constexpr int otile = kStages * kStageBytes;
    static constexpr int affine = otile + NW * kOWave;               // scale[256], shift[256]
    static constexpr int norm = affine + 2 * kD * 4;                 // [2][32] by tile parity: fg_scale / ||g||
    static constexpr int cshift = norm + 2 * kTilePx * 4;            // [32 NW]
    static constexpr int amax = cshift + 32 * NW * 4;                // [2][NW][32] float2
    static constexpr int total = amax + 2 * NW * kTilePx * 8;
};
using DecHl32Lds = DecHl32LdsT<4>;
static_assert(2 * DecHl32Lds::total <= 160 * 1024 && DecHl32LdsT<8, 3>::total <= 160 * 1024, "two workgroups of four waves / one of eight per CU");

// NSTG stages = NSTG - 1 tiles requested ahead: 2 for the four-wave form (two workgroups per CU hide each other's landing waits); the
// eight-wave form (one workgroup per CU) measured the same with 2 and 3 (SVPS_K2_HL32_W8_STAGES)
template <bool ARGMAX, int ABL = 0, int NW = 4, int NSTG = 2>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void mask_decode_hl32_kernel(
    const _Float16* __restrict__ feat_hi, const _Float16* __restrict__ feat_lo, const float* __restrict__ embed,
    const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, float fg_scale, float fg_shift, float* __restrict__ out,
    uint8_t* __restrict__ slot_argmax, int L, int HW, int tiles_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using Lds = DecHl32LdsT<NW, NSTG>;
    constexpr int NT = 64 * NW, NST = Lds::kStages, DST = NST - 1;     // DST: tiles requested ahead
    constexpr int PCW = 32 / NW;                                   // 1-KiB DMA pieces per wave and tile
    constexpr int HWV = NW / 2;                                    // waves per plane
    constexpr int TP = NT / kTilePx;                               // the norm's threads per pixel (8 or 16)
    constexpr int kMS = 4;                                         // mask stores per wave and tile
    constexpr int kAS = ARGMAX ? 1 : 0;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int t = blockIdx.y, c = blockIdx.x;
    const int px_begin = c * tiles_per_chunk * kTilePx;
    int px_end = px_begin + tiles_per_chunk * kTilePx;
    px_end = px_end < HW ? px_end : HW;
    const int nt = (px_end - px_begin + kTilePx - 1) / kTilePx;

    float* aff = reinterpret_cast<float*>(smem + Lds::affine);
    float* inv_norm = reinterpret_cast<float*>(smem + Lds::norm);
    float* cs = reinterpret_cast<float*>(smem + Lds::cshift);
    float2* am = reinterpret_cast<float2*>(smem + Lds::amax);
    for (int i = tid; i < kD; i += NT) {
        aff[i] = bn_scale[i];
        aff[kD + i] = bn_shift[i];
    }
    __syncthreads();

    f16x8 eh[16], el[16];                                          // (e . scale) of slot 32 w + r as fp16 hi + lo, A fragments
    {
        const int slot = 32 * w + r;
        const float* erow = embed + ((size_t)t * L + (slot < L ? slot : 0)) * kD + 8 * h;
        float dot = 0.f;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(erow + 16 * ks);
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(erow + 16 * ks + 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ch = 16 * ks + 8 * h + j;
                float x = j < 4 ? x0[j] : x1[j - 4];
                if (slot >= L) x = 0.f;
                dot += x * aff[kD + ch];
                float xs = x * aff[ch];
                asm volatile("" : "+v"(xs));                       // ONE fp32 value for both halves (see mask_decode_kernel_v2)
                const _Float16 hi = (_Float16)xs;
                eh[ks][j] = hi;
                el[ks][j] = (_Float16)(xs - (float)hi);
            }
            if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        dot = wave_half_xor_sum(dot);
        // a padded slot row (>= L) must never win the argmax: its constant is -inf x sign(fg_scale), so its logit is -inf (NaN for fg_scale = 0:
        // never greater than anything); its logits are not stored
        if (h == 0) cs[slot] = slot < L ? dot : (fg_scale >= 0.f ? -INFINITY : INFINITY);
    }
    wait_vm<0>();
    __syncthreads();
    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>((SVPS_LDS const void*)smem);
    const u32x4 frs = make_srd_d((w >= HWV ? feat_lo : feat_hi) + (size_t)t * HW * kD, (uint32_t)HW * kRowBytes);   // first half of the waves: hi plane
    const u32x4 ors = make_srd_d(out + (size_t)t * L * HW, (uint32_t)L * (uint32_t)HW * 4u);
    const u32x4 ars = make_srd_d(ARGMAX ? slot_argmax + (size_t)t * HW : nullptr, ARGMAX ? (uint32_t)HW : 0u);
    // wave w stages rows row0 .. row0 + 2 PCW - 1 of its plane (row0 = 2 PCW (w mod HWV)): PCW pieces of two rows, instruction groups of four
    // LDS position c of row `row` holds the logical chunk c ^ swz(row); with row = row0 + 8 g + 2 i + hh the swizzle is
    // ((2 (i & 1) + hh) << 2) | ((row0 / 4 + 2 g + (i >> 1)) & 3): a lane constant (rr ^ (hh << 2)) xor a wave-uniform constant per piece
    const int row0 = 2 * PCW * (w % HWV);
    auto stage = [&](int tile) {
        if (tile >= nt || (ABL & 8)) return;
        int rr = r, hh = h;
        asm volatile("" : "+v"(rr), "+v"(hh));
        const uint32_t st = __builtin_amdgcn_readfirstlane(lds0 + Lds::ring + (tile % NST) * Lds::kStageBytes + (w / HWV) * kTileBytes + row0 * kRowBytes);
        const int px0 = px_begin + tile * kTilePx;
        const int soff = __builtin_amdgcn_readfirstlane(px0 * kRowBytes);
        if (px0 + kTilePx <= HW) {
            const int cbase = (rr ^ (hh << 2)) * 16;                        // chunk term of this lane, before the piece's constant
            const int rbase = (row0 + hh) * kRowBytes;                      // row term
#pragma unroll
            for (int g = 0; g < PCW / 4; ++g) {
                int vo[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int kc = (((2 * (i & 1)) << 2) | (((row0 >> 2) + 2 * g + (i >> 1)) & 3)) * 16;
                    vo[i] = (cbase ^ kc) + rbase + (8 * g + 2 * i) * kRowBytes - 1024 * i;
                }
                dma16x4_d(frs, st + 4096 * g, vo[0], vo[1], vo[2], vo[3], soff);
            }
        } else {
#pragma unroll
            for (int i = 0; i < PCW; ++i) {
                const int row = row0 + 2 * i + hh;
                const int src = px0 + row < HW ? row : HW - 1 - px0;
                dma16_d(frs, st + i * 1024, src * kRowBytes + ((rr ^ swz(row)) * 16), soff);
            }
        }
    };
#pragma unroll
    for (int d = 0; d < DST; ++d) stage(d);

    // fragment addresses: chunk (2 ks + h) ^ swz(r) for ks = 0 .. 7; ks + 8 is + 256 B, the lo tile + kTileBytes (immediates)
    int fa[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) fa[k] = r * kRowBytes + (((2 * k + h) ^ swz(r)) * 16);
    char* ot = smem + Lds::otile + w * Lds::kOWave;
    // the norm's thread -> (pixel tid / TP, chunks (tid mod TP) + TP i): LDS offset of chunk i = 0 and first channel
    const int nrm_o0 = (tid / TP) * kRowBytes + (((tid % TP) ^ swz(tid / TP)) * 16);
    const int nrm_ch = 8 * (tid % TP);
    f32x16 sp;                                                     // accumulators of the previous tile
#pragma unroll
    for (int i = 0; i < 16; ++i) sp[i] = 0.f;

    auto body = [&](const int it, auto has_chain, auto has_epi) {
        constexpr bool CH = decltype(has_chain)::value, EP = decltype(has_epi)::value && !(ABL & 4);
        // issue order per iteration: DMA(it + DST) x PCW, argmax store of tile it - 2, mask stores of tile it - 1 x 4. Younger than DMA(it)
        // (requested in iteration it - DST): the stores of iterations it - DST .. it - 1 (if they had an epilogue) and the requests of tiles
        // it + 1 .. it + DST - 1 that exist
        if constexpr (CH) {
            if constexpr (DST == 1) {
                if (it >= 2) wait_vm<kMS + kAS>();
                else wait_vm<0>();
            } else {
                int younger = nt - 1 - it;
                younger = younger < DST - 1 ? younger : DST - 1;
                if (it >= DST + 1) wait_vm_dyn(DST * (kMS + kAS) + PCW * younger);
                else wait_vm<0>();
            }
        }
        wg_barrier();
        if constexpr (CH) stage(it + DST);
        const char* fth = smem + Lds::ring + (it % NST) * Lds::kStageBytes;

        if constexpr (CH) {   // fg_scale / ||scale (f_hi + f_lo) + shift|| per pixel: TP = 8 (16) threads per pixel, 4 (2) chunks each
            // TP = 8: chunk nsub + 8 i of pixel npx sits at LDS chunk (nsub + 8 i) ^ swz(npx): o_0 for i = 0, o_0 ^ 128 B for i = 1, + 256 B for i + 2
            // TP = 16: chunk nsub + 16 i: o_0 + 256 B i
            int o0 = nrm_o0;
            asm volatile("" : "+v"(o0));                                     // (opaque per tile: no address table in registers)
            const int o1 = o0 ^ 128;
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < 32 / TP; ++i) {
                const int o = TP == 8 ? ((i & 1) ? o1 : o0) + (i >> 1) * 256 : o0 + i * 256;
                const f16x8 xh = *reinterpret_cast<const f16x8*>(fth + o);
                const f16x8 xl = *reinterpret_cast<const f16x8*>(fth + kTileBytes + o);
                const float* ap = aff + nrm_ch + 8 * TP * i;                  // channels 8 (nsub + TP i) ..
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(ap), a1 = *reinterpret_cast<const f32x4*>(ap + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(ap + kD), b1 = *reinterpret_cast<const f32x4*>(ap + kD + 4);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float a_ = j < 4 ? a0[j] : a1[j - 4], b_ = j < 4 ? b0[j] : b1[j - 4];
                    // scale (hi + lo) + shift as two fused multiply-adds on the fp16 values (v_fma_mix_f32: no conversion instructions)
                    const float g = __builtin_fmaf((float)xl[j], a_, __builtin_fmaf((float)xh[j], a_, b_));
                    ss = __builtin_fmaf(g, g, ss);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            ss += __shfl_xor(ss, 1);
            ss += __shfl_xor(ss, 2);
            ss += __shfl_xor(ss, 4);
            if constexpr (TP == 16) ss += __shfl_xor(ss, 8);
            if ((tid % TP) == 0) inv_norm[(it & 1) * kTilePx + (tid / TP)] = fg_scale / fmaxf(sqrtf(ss), 1e-12f);
        }

        constexpr bool FR = CH && !(ABL & 2);
        constexpr int FAH = SVPS_K2_HL32_AHEAD, FNB = FAH + 1;
        f16x8 fh[FNB], fl[FNB];                          // operand fragments of one k-step, the next FAH k-steps' in flight
        if constexpr (FR) {
#pragma unroll
            for (int k = 0; k < FAH; ++k) {
                fh[k] = *reinterpret_cast<const f16x8*>(fth + fa[k & 7] + (k >> 3) * 256);
                fl[k] = *reinterpret_cast<const f16x8*>(fth + kTileBytes + fa[k & 7] + (k >> 3) * 256);
            }
        }
        float2 cnd[NW];                                  // four waves: requested here, used under k-step 1; eight: read there (registers)
        if constexpr (ARGMAX && EP && NW == 4) {
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) cnd[ww] = am[(it & 1) * NW * kTilePx + ww * kTilePx + r];
        }
        float inr = 0.f;
        f32x4 c4[2];                                     // e . shift of slots acc_row(4 q .. 4 q + 3, h), one quad ahead
        if constexpr (EP) {
            inr = inv_norm[((it - 1) & 1) * kTilePx + r];
            c4[0] = *reinterpret_cast<const f32x4*>(cs + 32 * w + 4 * h);
        }
        const int px_prev = px_begin + (it - 1) * kTilePx;
        auto half_store = [&](int hf) {        // slots 16 hf .. + 15 of this wave's block: two 1-KiB wave stores of 128-byte row segments
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int sl = 8 * u + (lane >> 3), cc = lane & 7;
                const u32x4 val = *reinterpret_cast<const u32x4*>(ot + sl * Lds::kORow + cc * 16);
                const int slot = 32 * w + 16 * hf + sl, px = px_prev + 4 * cc;
                const bool ok = slot < L && px < px_end && !(ABL & 64);
                store16_d(val, ors, ok ? (slot * HW + px) * 4 : 0x7ffffff0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
        float best = -INFINITY;
        int best_slot = 0x7fffffff;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if constexpr (EP) {
                if ((g & 1) == 0 && g < 6) c4[((g >> 1) + 1) & 1] = *reinterpret_cast<const f32x4*>(cs + 32 * w + 8 * ((g >> 1) + 1) + 4 * h);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ks = 2 * g + u;
                if constexpr (FR) {
                    if (ks + FAH < 16) {
                        fh[(ks + FAH) % FNB] = *reinterpret_cast<const f16x8*>(fth + fa[(ks + FAH) & 7] + ((ks + FAH) >> 3) * 256);
                        fl[(ks + FAH) % FNB] = *reinterpret_cast<const f16x8*>(fth + kTileBytes + fa[(ks + FAH) & 7] + ((ks + FAH) >> 3) * 256);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (ABL & 1) {
                        asm volatile("" : : "v"(fh[ks % FNB]), "v"(fl[ks % FNB]), "v"(el[ks]), "v"(eh[ks]));
                    } else {
                        s = mfma16(el[ks], fh[ks % FNB], s);
                        s = mfma16(eh[ks], fl[ks % FNB], s);
                        s = mfma16(eh[ks], fh[ks % FNB], s);
                    }
                }
                if constexpr (ARGMAX && EP) {
                    if (ks == 1) {                       // cross-wave part of the argmax of tile it - 2 (every wave computes, wave 0 stores)
                        float b = -INFINITY;
                        int bs = 0x7fffffff;
#pragma unroll
                        for (int ww = 0; ww < NW; ++ww) {
                            if constexpr (NW != 4) cnd[ww] = am[(it & 1) * NW * kTilePx + ww * kTilePx + r];
                            const int sl = __float_as_int(cnd[ww].y);
                            if (cnd[ww].x > b || (cnd[ww].x == b && sl < bs)) { b = cnd[ww].x; bs = sl; }
                        }
                        const int px = px_begin + (it - 2) * kTilePx + r;
                        const bool ok = w == 0 && h == 0 && it >= 2 && px < px_end;
                        store1_d(bs, ars, ok ? px : 0x7ffffff0);
                    }
                }
                if constexpr (EP) {
                    const int i = ks;                    // accumulator element = slot acc_row(i, h) of this wave's block
                    const int sl = acc_row(i, h);
                    const float m = (sp[i] + c4[(i >> 2) & 1][i & 3]) * inr + fg_shift;
                    *reinterpret_cast<float*>(ot + (sl & 15) * Lds::kORow + r * 4) = m;
                    if constexpr (ARGMAX) {
                        if (m > best) { best = m; best_slot = 32 * w + sl; }      // slots ascend with i within a lane; padded rows are -inf / NaN
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (EP) {
                if (g == 3) half_store(0);               // elements 0 .. 7 = slots 0 .. 15 are in the transpose tile
            }
        }
        if constexpr (EP) {
            half_store(1);
            if constexpr (ARGMAX) {
                const auto pb = __builtin_amdgcn_permlane32_swap(__float_as_uint(best), __float_as_uint(best), false, false);
                const auto ps = __builtin_amdgcn_permlane32_swap((uint32_t)best_slot, (uint32_t)best_slot, false, false);
                best = __uint_as_float(pb[0]);
                best_slot = (int)ps[0];
                const float ob = __uint_as_float(pb[1]);
                const int os = (int)ps[1];
                if (ob > best || (ob == best && os < best_slot)) { best = ob; best_slot = os; }
                if (h == 0) am[((it - 1) & 1) * NW * kTilePx + w * kTilePx + r] = make_float2(best, __int_as_float(best_slot));
            }
        }
        if constexpr (CH) sp = s;
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    body(0, T_{}, F_{});
#pragma clang loop unroll(disable)
    for (int it = 1; it < nt; ++it) body(it, T_{}, T_{});
    body(nt, F_{}, T_{});
    if constexpr (ARGMAX) {
        wg_barrier();
        if (w == 0 && h == 0) {                          // cross-wave part of the argmax of the last tile
            float b = -INFINITY;
            int bs = 0x7fffffff;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) {
                const float2 cn = am[((nt - 1) & 1) * NW * kTilePx + ww * kTilePx + r];
                const int sl = __float_as_int(cn.y);
                if (cn.x > b || (cn.x == b && sl < bs)) { b = cn.x; bs = sl; }
            }
            const int px = px_begin + (nt - 1) * kTilePx + r;
            if (px < px_end) slot_argmax[(size_t)t * HW + px] = (uint8_t)bs;
        }
    }
}

}  // namespace svps

namespace {

int dec_num_cus() { return svps_num_cus(); }

template <int NW, int NST, bool ARGMAX, typename OutT, typename MT = __bf16, bool HL = false>
hipError_t launch_decode(const void* feat, const float* embed, const float* bn_scale, const float* bn_shift,
                         float fg_scale, float fg_shift, void* out, uint8_t* slot_argmax, int T, int L,
                         int HW, hipStream_t stream, const void* feat_lo = nullptr) {
    using Lds = svps::DecLds<NW, NST, HL>;
    auto kern = svps::mask_decode_kernel<NW, NST, ARGMAX, OutT, MT, HL>;
    static SvpsLdsAttr attr;
    if (hipError_t ae = attr.ensure(reinterpret_cast<const void*>(kern), Lds::total); ae != hipSuccess) return ae;
    // sized for two workgroups per CU by LDS (2 x ~70 KiB); this first kernel needs 297 registers per lane, so in practice
    // one is resident - the fast path below is the one built for two
    const int tiles = (HW + svps::kTilePx - 1) / svps::kTilePx;
    int chunks = svps_pick_chunks(T, tiles, 2 * dec_num_cus());
    const int tpc = (tiles + chunks - 1) / chunks;
    chunks = (tiles + tpc - 1) / tpc;
    hipLaunchKernelGGL(kern, dim3(chunks, T), dim3(NW * 64), Lds::total, stream,
                       static_cast<const MT*>(feat), embed, bn_scale, bn_shift, fg_scale, fg_shift,
                       static_cast<OutT*>(out), slot_argmax, L, HW, tpc, static_cast<const MT*>(feat_lo));
    return hipGetLastError();
}

template <bool ARGMAX, int NW, bool LOGITS = true, int ABL = 0, int NSTG = 0, typename MT = __bf16, bool HL = false>
hipError_t launch_decode_v2(const void* feat, const float* embed, const float* bn_scale, const float* bn_shift,
                            float fg_scale, float fg_shift, void* out, uint8_t* slot_argmax, int T, int L, int HW,
                            hipStream_t stream, const void* feat_lo = nullptr) {
    auto kern = svps::mask_decode_kernel_v2<ARGMAX, NW, LOGITS, ABL, NSTG, MT, HL>;
    using Lds = svps::Dec2Lds<NW, LOGITS, NSTG>;
    static SvpsLdsAttr attr;
    if (hipError_t ae = attr.ensure(reinterpret_cast<const void*>(kern), Lds::total); ae != hipSuccess) return ae;
    constexpr int tpx = HL ? 16 : svps::kTilePx;                     // HL: a tile is sixteen pixels (hi rows + lo rows)
    const int tiles = (HW + tpx - 1) / tpx;                          // NW = 4: two co-resident workgroups per CU (2 x 68 KiB LDS); 8: one
    int chunks = svps_pick_chunks(T, tiles, (NW == 4 ? 2 : 1) * dec_num_cus());
    const int tpc = (tiles + chunks - 1) / chunks;
    chunks = (tiles + tpc - 1) / tpc;
    hipLaunchKernelGGL(kern, dim3(chunks, T), dim3(64 * NW), Lds::total, stream, static_cast<const MT*>(feat),
                       embed, bn_scale, bn_shift, fg_scale, fg_shift, static_cast<float*>(out), slot_argmax, L, HW, tpc,
                       static_cast<const MT*>(feat_lo));
    return hipGetLastError();
}

template <bool ARGMAX, int ABL = 0, int NW = 4, int NSTG = 2>
hipError_t launch_decode_hl32(const void* feat_hi, const void* feat_lo, const float* embed, const float* bn_scale, const float* bn_shift,
                              float fg_scale, float fg_shift, float* out, uint8_t* slot_argmax, int T, int L, int HW, hipStream_t stream) {
    auto kern = svps::mask_decode_hl32_kernel<ARGMAX, ABL, NW, NSTG>;
    using Lds = svps::DecHl32LdsT<NW, NSTG>;
    static SvpsLdsAttr attr;
    if (hipError_t ae = attr.ensure(reinterpret_cast<const void*>(kern), Lds::total); ae != hipSuccess) return ae;
    const int tiles = (HW + svps::kTilePx - 1) / svps::kTilePx;
    int chunks = svps_pick_chunks(T, tiles, (NW == 4 ? 2 : 1) * dec_num_cus());       // four waves: two co-resident workgroups per CU
    const int tpc = (tiles + chunks - 1) / chunks;
    chunks = (tiles + tpc - 1) / tpc;
    hipLaunchKernelGGL(kern, dim3(chunks, T), dim3(64 * NW), Lds::total, stream, static_cast<const _Float16*>(feat_hi),
                       static_cast<const _Float16*>(feat_lo), embed, bn_scale, bn_shift, fg_scale, fg_shift, out, slot_argmax, L, HW, tpc);
    return hipGetLastError();
}

template <int NW, int NST>
hipError_t dispatch_decode(const void* feat, const float* embed, const float* bn_scale, const float* bn_shift,
                           float fg_scale, float fg_shift, void* out, uint8_t* slot_argmax, int T, int L,
                           int HW, int flags, hipStream_t stream) {
    const bool bf = flags & SVPS_FLAG_OUT_BF16;
    if (slot_argmax)
        return bf ? launch_decode<NW, NST, true, __bf16>(feat, embed, bn_scale, bn_shift, fg_scale, fg_shift, out, slot_argmax, T, L, HW, stream)
                  : launch_decode<NW, NST, true, float>(feat, embed, bn_scale, bn_shift, fg_scale, fg_shift, out, slot_argmax, T, L, HW, stream);
    return bf ? launch_decode<NW, NST, false, __bf16>(feat, embed, bn_scale, bn_shift, fg_scale, fg_shift, out, slot_argmax, T, L, HW, stream)
              : launch_decode<NW, NST, false, float>(feat, embed, bn_scale, bn_shift, fg_scale, fg_shift, out, slot_argmax, T, L, HW, stream);
}

}  // namespace

#ifdef SVPS_K2_STAMP
extern "C" int svps_k2_debug_read(unsigned long long* stamps) {
    return (int)hipMemcpyFromSymbol(stamps, HIP_SYMBOL(svps::k2_stamps), sizeof(unsigned long long) * 4 * 8 * 8);
}
#endif

extern "C" int svps_mask_decode_fwd(const void* feat, const float* embed, const float* bn_scale,
                                    const float* bn_shift, float fg_scale, float fg_shift, void* out,
                                    uint8_t* slot_argmax, int T, int L, int HW, int D, int flags,
                                    void* stream_) {
    if (!feat || !embed || !bn_scale || !bn_shift || (!out && !slot_argmax)) return SVPS_ERR_BAD_ARG;
    if (D != svps::kD || T <= 0 || L <= 0 || L > 256 || HW <= 0) return SVPS_ERR_BAD_SHAPE;
    if ((size_t)HW > svps::kMaxFramePixels) return SVPS_ERR_BAD_SHAPE;       // 32-bit buffer offsets inside a frame
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    svps_prof_mark(SVPS_KERNEL_MASK_DECODE, 0, stream);
    if (flags & SVPS_FLAG_MAP_F16) {                       // fp16 fused map: fp32 logits (fast kernel, or the first one for ragged HW) / argmax only
        if ((flags & SVPS_FLAG_OUT_BF16) || (size_t)L * HW * 4 >= 0x7ffffff0u) return SVPS_ERR_BAD_ARG;
        using H = _Float16;
        hipError_t eh;
#define SVPS_H2(AM, W, LG) launch_decode_v2<AM, W, LG, 0, 0, H>(feat, embed, bn_scale, bn_shift, fg_scale, fg_shift, out, slot_argmax, T, L, HW, stream)
#define SVPS_H1(W, AM) launch_decode<W, 4, AM, float, H>(feat, embed, bn_scale, bn_shift, fg_scale, fg_shift, out, slot_argmax, T, L, HW, stream)
        if (!out) {
            if (HW & 3) return SVPS_ERR_BAD_SHAPE;
            eh = L <= 128 ? SVPS_H2(true, 4, false) : SVPS_H2(true, 8, false);
        } else if ((HW & 3) == 0) {
            eh = L <= 128 ? (slot_argmax ? SVPS_H2(true, 4, true) : SVPS_H2(false, 4, true))
                          : (slot_argmax ? SVPS_H2(true, 8, true) : SVPS_H2(false, 8, true));
        } else {
            eh = L <= 128 ? (slot_argmax ? SVPS_H1(4, true) : SVPS_H1(4, false)) : (slot_argmax ? SVPS_H1(8, true) : SVPS_H1(8, false));
        }
#undef SVPS_H2
#undef SVPS_H1
        svps_prof_mark(SVPS_KERNEL_MASK_DECODE, 1, stream);
        return (int)eh;
    }
    // fast path: 16-byte row-segment stores need 4-pixel alignment of every slot row; L * HW * 4 must fit a buffer descriptor
    static const bool legacy = getenv("SVPS_K2_LEGACY") != nullptr;   // comparison runs: the first-generation kernel
    const bool fast = (HW & 3) == 0 && !(flags & SVPS_FLAG_OUT_BF16) && (size_t)L * HW * 4 < 0x7ffffff0u && !legacy;
    if (!out) {                                          // argmax-only mode: the fast kernel without its logit stores
        if ((flags & SVPS_FLAG_OUT_BF16) || legacy) return SVPS_ERR_BAD_ARG;
#ifdef SVPS_K2_ABLATE
        if (const char* ae = getenv("SVPS_K2_ABLATE"); ae && L <= 128) {
#define SVPS_K2A(N) case N: ea = launch_decode_v2<true, 4, false, N>(feat, embed, bn_scale, bn_shift, fg_scale, fg_shift, nullptr, slot_argmax, T, L, HW, stream); break;
            hipError_t ea = hipErrorUnknown;
            switch (atoi(ae)) {
                SVPS_K2A(1) SVPS_K2A(2) SVPS_K2A(4) SVPS_K2A(5) SVPS_K2A(6) SVPS_K2A(8) SVPS_K2A(14) SVPS_K2A(32)
                case 16: ea = launch_decode_v2<true, 4, false, 0, 3>(feat, embed, bn_scale, bn_shift, fg_scale, fg_shift, nullptr, slot_argmax, T, L, HW, stream); break;
                default: break;
            }
#undef SVPS_K2A
            if (ea != hipErrorUnknown) {
                svps_prof_mark(SVPS_KERNEL_MASK_DECODE, 1, stream);
                return (int)ea;
            }
        }
#endif
        hipError_t e0 = L <= 128 ? launch_decode_v2<true, 4, false>(feat, embed, bn_scale, bn_shift, fg_scale, fg_shift, nullptr, slot_argmax, T, L, HW, stream)
                                 : launch_decode_v2<true, 8, false>(feat, embed, bn_scale, bn_shift, fg_scale, fg_shift, nullptr, slot_argmax, T, L, HW, stream);
        svps_prof_mark(SVPS_KERNEL_MASK_DECODE, 1, stream);
        return (int)e0;
    }
#define SVPS_V2(AM, W) launch_decode_v2<AM, W>(feat, embed, bn_scale, bn_shift, fg_scale, fg_shift, out, slot_argmax, T, L, HW, stream)
    hipError_t e = fast ? (L <= 128 ? (slot_argmax ? SVPS_V2(true, 4) : SVPS_V2(false, 4))
                                    : (slot_argmax ? SVPS_V2(true, 8) : SVPS_V2(false, 8)))
                   : L <= 128 ? dispatch_decode<4, 4>(feat, embed, bn_scale, bn_shift, fg_scale, fg_shift, out,
                                                    slot_argmax, T, L, HW, flags, stream)
                            : dispatch_decode<8, 4>(feat, embed, bn_scale, bn_shift, fg_scale, fg_shift, out,
                                                    slot_argmax, T, L, HW, flags, stream);
#undef SVPS_V2
    svps_prof_mark(SVPS_KERNEL_MASK_DECODE, 1, stream);
    return (int)e;
}

// The decode of the reference-precision mode: the finest fused map as fp16 hi + lo planes (svps_level_fuse_hl_fwd), fp32 logits
// [T, L, HW] and / or the fused per-pixel slot argmax; the first-generation kernel (any HW, L <= 256) with both planes staged.
extern "C" int svps_mask_decode_hl_fwd(const void* feat_hi, const void* feat_lo, const float* embed, const float* bn_scale,
                                       const float* bn_shift, float fg_scale, float fg_shift, float* out, uint8_t* slot_argmax,
                                       int T, int L, int HW, int D, void* stream_) {
    if (!feat_hi || !feat_lo || !embed || !bn_scale || !bn_shift || !out) return SVPS_ERR_BAD_ARG;
    if (D != svps::kD || T <= 0 || L <= 0 || L > 256 || HW <= 0) return SVPS_ERR_BAD_SHAPE;
    if ((size_t)HW > svps::kMaxFramePixels) return SVPS_ERR_BAD_SHAPE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    using H = _Float16;
    svps_prof_mark(SVPS_KERNEL_MASK_DECODE, 0, stream);
    // L <= 128: four waves, two feature tiles per plane in flight, TWO workgroups per CU (2 x 68 KiB of LDS, 256 registers): this first-generation
    // kernel waits for each tile (hipcc drains its builtin LDS-DMA before the next LDS read), so a second workgroup is what hides the latency
#define SVPS_HL(W, NS, AM) launch_decode<W, NS, AM, float, H, true>(feat_hi, embed, bn_scale, bn_shift, fg_scale, fg_shift, out, slot_argmax, T, L, HW, stream, feat_lo)
    hipError_t e;
    static const bool old_form = getenv("SVPS_K2_HL_V1") != nullptr;         // comparison runs only (tools/kbench3.py)
#ifdef SVPS_K2_ABLATE
    if (const char* ae = getenv("SVPS_K2_ABLATE"); ae && L <= 128 && (HW & 3) == 0 && slot_argmax) {      // timing-only (tools/kbench_k2hl.py)
#define SVPS_K2A(N) case N: ea = launch_decode_v2<true, 4, true, N, 0, H, true>(feat_hi, embed, bn_scale, bn_shift, fg_scale, fg_shift, out, slot_argmax, T, L, HW, stream, feat_lo); break;
        hipError_t ea = hipErrorUnknown;
        switch (atoi(ae)) {
            SVPS_K2A(1) SVPS_K2A(2) SVPS_K2A(4) SVPS_K2A(5) SVPS_K2A(6) SVPS_K2A(8) SVPS_K2A(14) SVPS_K2A(64) SVPS_K2A(65) SVPS_K2A(72)
            default: break;
        }
#undef SVPS_K2A
        if (ea != hipErrorUnknown) {
            svps_prof_mark(SVPS_KERNEL_MASK_DECODE, 1, stream);
            return (int)ea;
        }
    }
#endif
    static const bool hl16 = getenv("SVPS_K2_HL16") != nullptr;              // comparison runs only: round 5's 16-pixel hi / lo tiles
    if (L <= 128 && (HW & 3) == 0 && !old_form && !hl16 && (size_t)L * HW * 4 < 0x7ffffff0u) {
        // round 6: 32-pixel tiles, three MFMAs per k-step into one accumulator (mask_decode_hl32_kernel)
        e = slot_argmax ? launch_decode_hl32<true>(feat_hi, feat_lo, embed, bn_scale, bn_shift, fg_scale, fg_shift, out, slot_argmax, T, L, HW, stream)
                        : launch_decode_hl32<false>(feat_hi, feat_lo, embed, bn_scale, bn_shift, fg_scale, fg_shift, out, slot_argmax, T, L, HW, stream);
    } else if (L > 128 && (HW & 3) == 0 && !old_form && !hl16 && (size_t)L * HW * 4 < 0x7ffffff0u) {
        // the same kernel with eight waves (up to 256 slots: VIPER's 200), one workgroup per CU
        e = slot_argmax ? launch_decode_hl32<true, 0, 8, SVPS_K2_HL32_W8_STAGES>(feat_hi, feat_lo, embed, bn_scale, bn_shift, fg_scale, fg_shift, out, slot_argmax, T, L, HW, stream)
                        : launch_decode_hl32<false, 0, 8, SVPS_K2_HL32_W8_STAGES>(feat_hi, feat_lo, embed, bn_scale, bn_shift, fg_scale, fg_shift, out, slot_argmax, T, L, HW, stream);
    } else if (L <= 128 && (HW & 3) == 0 && !old_form) {
        // round 5: the skewed fast path on 16-pixel hi / lo tiles (mask_decode_kernel_v2<.., HL>)
        e = slot_argmax ? launch_decode_v2<true, 4, true, 0, 0, H, true>(feat_hi, embed, bn_scale, bn_shift, fg_scale, fg_shift, out, slot_argmax, T, L, HW, stream, feat_lo)
                        : launch_decode_v2<false, 4, true, 0, 0, H, true>(feat_hi, embed, bn_scale, bn_shift, fg_scale, fg_shift, out, slot_argmax, T, L, HW, stream, feat_lo);
    } else {
        e = L <= 128 ? (slot_argmax ? SVPS_HL(4, 2, true) : SVPS_HL(4, 2, false)) : (slot_argmax ? SVPS_HL(8, 4, true) : SVPS_HL(8, 4, false));
    }
#undef SVPS_HL
    svps_prof_mark(SVPS_KERNEL_MASK_DECODE, 1, stream);
    return (int)e;
}
