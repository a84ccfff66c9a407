// Shared device helpers for the gfx950 (MI355X / CDNA4) slot-retriever kernels.
//
// Conventions used by every kernel in this directory
//   * wavefront = 64 lanes; lane l is split as r = l & 31 (MFMA row/col index) and h = l >> 5
//   * v_mfma_f32_32x32x16_bf16 operand maps (cdna_hip_programming.md section 3):
//       A[row r][k = 8h + j]   B[k = 8h + j][col r]   j = 0..7 (one bf16x8 per lane)
//       C/D: col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * h,  reg = 0..15
//   * pixel rows of a key/value/feature tile are 256 channels * 2 B = 512 B = 32 chunks of 16 B.
//     A tile is staged into LDS by LDS-DMA (global_load_lds_dwordx4, lane-linear destination); the
//     bank-conflict swizzle is therefore applied on the per-lane SOURCE address and again on every
//     read:  LDS chunk position = logical chunk ^ swz(row).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace svps {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define SVPS_LDS __attribute__((address_space(3)))
#define SVPS_GLB __attribute__((address_space(1)))

constexpr int kD = 256;              // channel width of slots / keys / values (dh_dim)
constexpr int kRowBytes = kD * 2;    // one bf16 pixel row
constexpr int kTilePx = 32;          // pixels per LDS tile (one MFMA column block)
constexpr int kTileBytes = kTilePx * kRowBytes;  // 16 KiB
constexpr size_t kMaxFramePixels = (size_t)1 << 22;   // 4 Mi pixels per frame and level: byte offsets inside a frame fit 31 bits
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kNegBig = -1.0e30f;

// 16-byte-chunk swizzle of a 512-B pixel row. Bijective on the low four chunk bits for any 16
// consecutive rows (ds_read_b128 of one chunk column over 16 rows is conflict-free) and spreads
// the 4 rows x 4 chunks of a ds_read_b64_tr_b16 half-wave over 16 distinct 16-B bank slots.
__device__ __forceinline__ int swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Workgroup barrier that does NOT drain the vector-memory counter (LDS-DMA stays in flight).
__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field on gfx9");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (0..31; larger values wait for everything)
__device__ __forceinline__ void wait_vm_dyn(int n) {
    switch (n) {
#define SVPS_WV(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        SVPS_WV(0) SVPS_WV(1) SVPS_WV(2) SVPS_WV(3) SVPS_WV(4) SVPS_WV(5) SVPS_WV(6) SVPS_WV(7) SVPS_WV(8) SVPS_WV(9)
        SVPS_WV(10) SVPS_WV(11) SVPS_WV(12) SVPS_WV(13) SVPS_WV(14) SVPS_WV(15) SVPS_WV(16) SVPS_WV(17) SVPS_WV(18)
        SVPS_WV(19) SVPS_WV(20) SVPS_WV(21) SVPS_WV(22) SVPS_WV(23) SVPS_WV(24) SVPS_WV(25) SVPS_WV(26) SVPS_WV(27)
        SVPS_WV(28) SVPS_WV(29) SVPS_WV(30) SVPS_WV(31)
#undef SVPS_WV
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// One 1-KiB LDS-DMA piece: 64 lanes x 16 B, destination = lds_base + lane * 16 (hardware adds the
// lane term; lds_base must be wave-uniform), source = per-lane global pointer.
__device__ __forceinline__ void dma16(const void* gsrc, char* lds_base) {
    __builtin_amdgcn_global_load_lds((const SVPS_GLB void*)gsrc, (SVPS_LDS void*)lds_base, 16, 0, 0);
}

// Stage `rows_per_wave`-row slice of a [kTilePx x 512 B] tile. Wave `w` of `NW` issues
// kTilePx / 2 / NW pieces, each covering two pixel rows. `px0` = first pixel of the tile,
// `px_last` = last readable pixel of the frame (rows past it are clamped: their contribution is
// masked by the caller). `base` points at pixel 0 of the frame.
template <int NW>
__device__ __forceinline__ void dma_tile(const char* base, int px0, int px_last, char* lds_tile,
                                         int w, int lane) {
    constexpr int PIECES = kTilePx / 2 / NW;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int j = w * PIECES + i;          // piece index: rows 2j, 2j + 1
        const int row = 2 * j + (lane >> 5);
        const int chunk = (lane & 31) ^ swz(row);
        int px = px0 + row;
        px = px < px_last ? px : px_last;
        dma16(base + (size_t)px * kRowBytes + chunk * 16, lds_tile + j * 1024);
    }
}

// B-operand fragment of X^T for S = Q * X^T (contraction over channels): lane (r, h) gets
// X[row r][channels 16 ks + 8 h .. + 8].
__device__ __forceinline__ bf16x8 read_row_frag(const char* lds_tile, int ks, int r, int h) {
    const int chunk = (2 * ks + h) ^ swz(r);
    return *reinterpret_cast<const bf16x8*>(lds_tile + r * kRowBytes + chunk * 16);
}
template <typename V>                                          // the same fragment as 8 x bf16 or 8 x fp16 (map element type)
__device__ __forceinline__ V read_row_frag_as(const char* lds_tile, int ks, int r, int h) {
    const int chunk = (2 * ks + h) ^ swz(r);
    return *reinterpret_cast<const V*>(lds_tile + r * kRowBytes + chunk * 16);
}

// B-operand fragment of V for O += P * V (contraction over pixels): lane (n, h) gets
// V[pixels 16 ks + 8 h .. + 8][channel 32 db + n], via two hardware-transposed LDS reads.
// ds_read_b64_tr_b16: per 16-lane group, lane 4q+p supplies the address of row q / columns
// 4p..4p+3 of a 4 x 16 block; lane i receives column i of the four rows.
__device__ __forceinline__ bf16x8 read_col_frag(const char* lds_tile, int ks, int db, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int row0 = 16 * ks + 8 * (g >> 1) + q;
    const int chunk = 4 * db + 2 * (g & 1) + (p >> 1);
    const int sub = 8 * (p & 1);
    const char* a0 = lds_tile + row0 * kRowBytes + ((chunk ^ swz(row0)) * 16) + sub;
    const char* a1 = lds_tile + (row0 + 4) * kRowBytes + ((chunk ^ swz(row0 + 4)) * 16) + sub;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((SVPS_LDS bf16x4*)a0);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((SVPS_LDS bf16x4*)a1);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ float wave_half_xor_max(float x) { return fmaxf(x, __shfl_xor(x, 32)); }
__device__ __forceinline__ float wave_half_xor_sum(float x) { return x + __shfl_xor(x, 32); }

// The fused level maps are 16-bit in HBM: bf16 (default, BASELINE's storage) or fp16 (MultiScaleDynamicMaskHead.map_dtype = "fp16":
// three more mantissa bits for the same bytes, |f| < 65 504). Kernels templated on the map's element type use these overloads.
__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ bf16x4 ds_tr16(SVPS_LDS bf16x4* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4bf16(p); }
typedef __fp16 svps_h4 __attribute__((ext_vector_type(4)));       // the builtin's own element type
__device__ __forceinline__ f16x4 ds_tr16(SVPS_LDS f16x4* p) {
    return __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((SVPS_LDS svps_h4*)p));
}

// The probabilities of the fused retriever travel as fp16 (P * rstd_v, csrc/retr_attn.hip): below 6.1e-5 fp16 is subnormal, below
// 6e-8 zero - a slot that owns almost no pixel (P ~ 1e-7 everywhere; common once the softmax over slots is sharp) lost its whole
// contribution, and norm1 behind the retriever scales such a row up by up to 1 / sqrt(eps). The kernels therefore carry
// 2^7 * P * rstd_v (exact: a power of two) and retr_finish_kernel takes the factor out again: seven more binades at the bottom.
// Upper end: P <= 1 and rstd_v <= 1 / sqrt(eps_v), so 2^7 * P * rstd_v <= 40 477 < 65 504 for eps_v >= 1e-5 (the host side refuses
// eps_v < 4e-6, slot_head.MaskDynamicConv.forward_fused).
constexpr float kPScale = 128.f;
constexpr float kPScaleInv = 1.f / 128.f;

// row (slot) index of accumulator register `reg` inside a 32 x 32 C/D tile
__device__ __forceinline__ constexpr int acc_row(int reg, int h) {
    return (reg & 3) + 8 * (reg >> 2) + 4 * h;
}

}  // namespace svps

// Host side, launch state that is per DEVICE, not per process (a process may drive several GPUs): the CU count and the
// "dynamic LDS above 64 KiB allowed" attribute of a kernel, both looked up for the device current at the call.
static inline int svps_cur_device() {
    int d = 0;
    return hipGetDevice(&d) == hipSuccess ? (d & 63) : 0;
}
static inline int svps_num_cus() {
    static int n[64];
    const int d = svps_cur_device();
    if (n[d] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) != hipSuccess) return 256;
        n[d] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return n[d];
}
// GELU, exact erf form (F.gelu's default). No contraction of its own products and sums: the value must not depend on the kernel
// the function is inlined into (svps_slot_ffn is tested bit for bit against svps_slot_gemm + svps_slot_gemm_ln).
__device__ __forceinline__ float svps_gelu_erf(float x) {
#pragma clang fp contract(off)
    const float e = erff(x * 0.70710678118654752f);
    const float t = 0.5f * x;
    return t * (1.0f + e);
}

struct SvpsLdsAttr {          // one static instance per launch site (= per kernel instantiation)
    bool done[64] = {};
    hipError_t ensure(const void* kernel, int lds_bytes) {
        const int d = svps_cur_device();
        if (done[d]) return hipSuccess;
        const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e == hipSuccess) done[d] = true;
        return e;
    }
};

// Host side: workgroups per frame for a launch of T frames on `slots` resident workgroups (CUs x workgroups per CU).
// All workgroups of a launch do the same amount of work, so the launch runs in whole "rounds" of `slots` workgroups:
// take the smallest count >= slots / T whose last round is (nearly) full - e.g. 80 frames on 256 CUs: 3 per frame would
// leave 16 CUs idle for the whole launch, 16 per frame fills five rounds exactly.
static inline int svps_pick_chunks(int T, int tiles, int slots, int min_tiles = 16, int max_mult = 8) {
    int base = slots / (T > 0 ? T : 1);
    if (base < 1) base = 1;
    if (base >= tiles) return tiles;
    int best = base;
    double best_eff = 0.0;
    int hi = base * max_mult < tiles ? base * max_mult : tiles;   // more workgroups = more per-workgroup prologue / partial traffic:
    if (hi > tiles / min_tiles) hi = tiles / min_tiles;           // keep at least `min_tiles` tiles per workgroup
    if (hi < base) hi = base;
    for (int c = base; c <= hi; ++c) {
        const long wg = (long)T * c;
        const long rounds = (wg + slots - 1) / slots;
        const double eff = (double)wg / (double)(rounds * slots);
        if (eff > best_eff + 1e-9) { best_eff = eff; best = c; }
        if (eff > 0.995) break;
    }
    return best;
}
