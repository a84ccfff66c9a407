// Hardware-semantics probes: one-wave kernels that exercise exactly the helper functions the
// production kernels use, so a wrong lane map shows up as a wrong matrix instead of a wrong mask.
#include "common.h"
#include "../../include/slotvps_hip.h"
#include "../../include/slotvps_hip_diag.h"

namespace svps {

// c[32,32] = a[32,16] @ b[16,32] with one v_mfma_f32_32x32x16_bf16
__global__ __launch_bounds__(64) void probe_mfma_kernel(const __bf16* a, const __bf16* b, float* c) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    bf16x8 af, bfr;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        af[j] = a[r * 16 + 8 * h + j];       // A[row r][k = 8h + j]
        bfr[j] = b[(8 * h + j) * 32 + r];    // B[k = 8h + j][col r]
    }
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) c[acc_row(i, h) * 32 + r] = acc[i];
}

// x[32,256] -> LDS through the swizzled LDS-DMA path, then read back both ways
__global__ __launch_bounds__(64) void probe_tile_kernel(const __bf16* x, __bf16* rows, __bf16* cols) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    dma_tile<1>(reinterpret_cast<const char*>(x), 0, kTilePx - 1, smem, 0, lane);
    wait_vm<0>();
    wg_barrier();
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const bf16x8 f = read_row_frag(smem, ks, r, h);
#pragma unroll
        for (int j = 0; j < 8; ++j) rows[r * kD + 16 * ks + 8 * h + j] = f[j];
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int db = 0; db < 8; ++db) {
            const bf16x8 f = read_col_frag(smem, ks, db, lane);
#pragma unroll
            for (int j = 0; j < 8; ++j) cols[(16 * ks + 8 * h + j) * kD + 32 * db + r] = f[j];
        }
}

// Known-bytes streaming kernel: 16 B per lane in, 16 B per lane out, every byte once - the access shape of the library's own
// streams. bench.py times it (a hand-written copy ceiling next to the vendor peak) and tools/pmc_traffic.py calibrates the
// FETCH_SIZE / WRITE_SIZE counters on it inside the profiled process.
__global__ __launch_bounds__(256) void probe_copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

// Mixed-traffic streaming probe: per unit (one 1-KiB wave instruction's worth per wave) RI KiB are read and RO KiB written, every
// byte once, nontemporal - the read : write mix of a kernel without its arithmetic (K4 with the reference's fp32 NCHW input moves
// 640 B in : 512 B out per pixel = 5 : 4; K2 with fp32 logits 512 : 401). The loaded values are folded into what is stored.
template <int RI, int RO>
__global__ __launch_bounds__(256) void probe_mix_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t units) {
    const size_t stride = (size_t)gridDim.x * 4;                      // units in flight: one per wave
    const int lane = threadIdx.x & 63;
    u32x4 keep = {0u, 0u, 0u, 0u};
    for (size_t u = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); u < units; u += stride) {
        u32x4 v[RI > 0 ? RI : 1];
#pragma unroll
        for (int i = 0; i < RI; ++i) v[i] = __builtin_nontemporal_load(src + (u * RI + i) * 64 + lane);
        u32x4 x = keep;
#pragma unroll
        for (int i = 0; i < RI; ++i) x ^= v[i];
#pragma unroll
        for (int i = 0; i < RO; ++i) __builtin_nontemporal_store(x, dst + (u * RO + i) * 64 + lane);
        if constexpr (RO == 0) keep = x;
    }
    if constexpr (RO == 0) {                                          // read-only: one store per lane at the end keeps the loads alive
        if (keep[0] == 0x9e3779b9u && keep[1] == 0x7f4a7c15u) dst[lane] = keep;
    }
}

}  // namespace svps

extern "C" int svps_probe_mix(const void* src, void* dst, size_t units, int ri, int ro, void* stream_) {
    if (!src || !dst) return SVPS_ERR_BAD_ARG;
    if (units == 0) return SVPS_ERR_BAD_SHAPE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const dim3 grid(svps_num_cus() * 8), block(256);
    const svps::u32x4* s = static_cast<const svps::u32x4*>(src);
    svps::u32x4* d = static_cast<svps::u32x4*>(dst);
#define SVPS_MIX(RI, RO) if (ri == RI && ro == RO) { hipLaunchKernelGGL((svps::probe_mix_kernel<RI, RO>), grid, block, 0, stream, s, d, units); return (int)hipGetLastError(); }
    SVPS_MIX(5, 4) SVPS_MIX(3, 4) SVPS_MIX(1, 1) SVPS_MIX(1, 0) SVPS_MIX(0, 1) SVPS_MIX(4, 1) SVPS_MIX(2, 1)
#undef SVPS_MIX
    return SVPS_ERR_BAD_SHAPE;
}

extern "C" int svps_probe_copy(const void* src, void* dst, size_t bytes, void* stream) {
    if (!src || !dst) return SVPS_ERR_BAD_ARG;
    if (bytes == 0 || (bytes & 15)) return SVPS_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(svps::probe_copy_kernel, dim3(svps_num_cus() * 8), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const svps::u32x4*>(src), static_cast<svps::u32x4*>(dst), bytes / 16);
    return (int)hipGetLastError();
}

extern "C" int svps_probe_mfma(const void* a, const void* b, float* c, void* stream) {
    if (!a || !b || !c) return SVPS_ERR_BAD_ARG;
    hipLaunchKernelGGL(svps::probe_mfma_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream),
                       static_cast<const __bf16*>(a), static_cast<const __bf16*>(b), c);
    return (int)hipGetLastError();
}

extern "C" int svps_probe_tile(const void* x, void* rows, void* cols, void* stream) {
    if (!x || !rows || !cols) return SVPS_ERR_BAD_ARG;
    hipLaunchKernelGGL(svps::probe_tile_kernel, dim3(1), dim3(64), svps::kTileBytes,
                       static_cast<hipStream_t>(stream), static_cast<const __bf16*>(x),
                       static_cast<__bf16*>(rows), static_cast<__bf16*>(cols));
    return (int)hipGetLastError();
}
