// Shared pieces of the statistics-fused retriever kernels (retr_attn.hip: 16-bit forms and the 16-pixel hi / lo form; retr_attn_hl32.hip:
// the 32-pixel hi / lo form of round 6): LDS tile sizes of the aux / P / Cy rings, the asm LDS-DMA forms, half-wave exchanges.
#pragma once
#include "common.h"

namespace svps {

constexpr int kAuxRow = 16;                // bytes per pixel of the aux tensor (retr_stats.hip)
constexpr int kAuxTile = 1024;             // LDS per staged aux tile: 512 B of rows (+ 512 B the upper half of the DMA instruction repeats)
constexpr int kPTile = 8192;               // P tile: 128 slots x 32 pixels fp16
constexpr int kCyTile = 1024;              // one LDS-DMA piece: the Cy row of the tile's image row (LP = 128: and the next row)
constexpr int kPartRow = 260;              // floats per slot row of a partial: 256 channels of A + 4 aux columns (every byte of a partial is written)
constexpr int kExtRow = 272;               // floats per slot row of the finished result: 17 k-steps of 16 for the slot-side product


__device__ __forceinline__ u32x4 ra_make_srd(const void* base, uint32_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    u32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
    d[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
    d[2] = __builtin_amdgcn_readfirstlane(bytes);
    d[3] = 0x00020000u;
    return d;
}

// asm LDS-DMA with the non-temporal hint (the map is read once per launch); see slot_attn.hip for why this is asm
__device__ __forceinline__ void ra_dma16(u32x4 srd, uint32_t lds_addr, int voff, int soff) {
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
// the same without the hint: the Cy rows are re-read by every workgroup of the frame (L2-resident)
__device__ __forceinline__ void ra_dma16_cached(u32x4 srd, uint32_t lds_addr, int voff) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %2, %3, 0 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_addr), "v"(voff), "s"(srd)
        : "memory");
}

__device__ __forceinline__ float ra_half_swap_max(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float ra_half_swap_sum(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// retr_attn_hl32.hip: K1'-HL on 32-pixel tiles (hi tile + lo tile per ring stage). Launches the kernel for the `L` (<= 128) slot rows
// starting at `slot_off` of an LP-row layout; ext_stats != nullptr: softmax statistics over ALL slots from retr_logit_stats_kernel.
// Returns a hipError_t. `tiles_per_chunk` counts 32-pixel tiles (retr_hl32_tile_px).
constexpr int kRetrHl32TilePx = 32;
int retr_attn_hl32_launch(const void* qh, const void* ql, const float* cy, const float* cx, const float* c3, const void* feat_hi,
                          const void* feat_lo, const void* aux, float* partial, int T, int L, int H, int W, int chunks, int tiles_per_chunk,
                          int LP, int Lrow, int slot_off, const void* ext_stats, void* stream);

// the logit statistics of the more-than-128-slot path on the same tiles ([T, HW] of (max, 1 / sum) over the 256 slot rows of the layout)
int retr_logit_stats_hl32_launch(const void* qh, const void* ql, const float* cy, const float* cx, const float* c3, const void* feat_hi,
                                 const void* feat_lo, const void* aux, void* stats, int T, int H, int W, int chunks, int tiles_per_chunk,
                                 void* stream);

}  // namespace svps
