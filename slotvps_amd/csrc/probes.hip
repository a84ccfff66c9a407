// Hardware-semantics probes: one-wave kernels that exercise exactly the helper functions the
// production kernels use, so a wrong lane map shows up as a wrong matrix instead of a wrong mask.
#include "common.h"
#include "../../include/slotvps_hip.h"

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

}  // namespace svps

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
