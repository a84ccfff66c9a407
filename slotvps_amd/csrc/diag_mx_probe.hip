// DIAGNOSTICS library: raw-register probe of the block-scaled matrix instruction v_mfma_scale_f32_32x32x64_f8f6f4 with FP8 (e4m3) operands
// (tools/mx_probe.py pins its A / B lane maps and the meaning of the scale operands with one-hot operands before any kernel relies on them).
//   a_regs, b_regs [N][64 lanes][8 dwords]: the operand registers of every lane, as given; scale_a, scale_b [N][64]: the scale VGPRs;
//   c_out [N][64][16] fp32: the result registers (accumulator started from zero). One wave per problem.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/slotvps_hip.h"
#include "../../include/slotvps_hip_diag.h"

namespace {
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(64) void mx_probe_kernel(const int* __restrict__ a_regs, const int* __restrict__ b_regs, const int* __restrict__ sa,
                                                      const int* __restrict__ sb, float* __restrict__ c_out) {
    const int n = blockIdx.x, lane = threadIdx.x;
    v8i a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = a_regs[((size_t)n * 64 + lane) * 8 + i];
        b[i] = b_regs[((size_t)n * 64 + lane) * 8 + i];
    }
    int s_a = sa[(size_t)n * 64 + lane], s_b = sb[(size_t)n * 64 + lane];
    asm volatile("" : "+v"(s_a), "+v"(s_b));               // registers, not literals (a literal scale is taken as an f32 constant)
    v16f c;
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0 /* A: fp8 e4m3 */, 0 /* B: fp8 e4m3 */, 0, s_a, 0, s_b);
#pragma unroll
    for (int i = 0; i < 16; ++i) c_out[((size_t)n * 64 + lane) * 16 + i] = c[i];
}
// out[i] = the two fp8 bytes v_cvt_scalef32_pk_fp8_f16 makes of the fp16 pair x[i] with `scale` (low half of the result dword)
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef short s2_t __attribute__((ext_vector_type(2)));
__global__ void cvt_probe_kernel(const h2_t* __restrict__ x, float scale, int* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    s2_t old = {0, 0};
    const s2_t r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(old, x[i], scale, false);
    out[i] = __builtin_bit_cast(int, r);
}
// The building block of the FP8 correction products (retr_stats_hl.hip, F8 form), end to end: a [32 rows][64 k], b [32 cols][64 k] as fp16;
// lane (r, h) takes its four "fp16 k-step fragments" (k = 16 q + 8 h + j, q = 0 .. 3, j = 0 .. 7), converts each with
// v_cvt_scalef32_pk_fp8_f16 (lo then hi half of two dwords) using the E8M0 bytes sa[lane] / sb[lane], and multiplies on the scaled MFMA:
// c [64 lanes][16] = a b^T within the FP8 rounding of the operands.
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void probe_cvt8(const h8_t& f, int sbyte, int& d0, int& d1) {
    const float sc = __uint_as_float((uint32_t)sbyte << 23);
    s2_t x = {0, 0}, y = {0, 0};
    x = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(x, h2_t{f[0], f[1]}, sc, false);
    x = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(x, h2_t{f[2], f[3]}, sc, true);
    y = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(y, h2_t{f[4], f[5]}, sc, false);
    y = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(y, h2_t{f[6], f[7]}, sc, true);
    d0 = __builtin_bit_cast(int, x);
    d1 = __builtin_bit_cast(int, y);
}
__global__ __launch_bounds__(64) void mx_block_kernel(const _Float16* __restrict__ a, const _Float16* __restrict__ b, const int* __restrict__ sa,
                                                      const int* __restrict__ sb, float* __restrict__ c_out, int* __restrict__ regs_out) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    v8i av, bv;
    int s_a = sa[lane], s_b = sb[lane];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const h8_t fa = *reinterpret_cast<const h8_t*>(a + r * 64 + 16 * q + 8 * h);
        const h8_t fb = *reinterpret_cast<const h8_t*>(b + r * 64 + 16 * q + 8 * h);
        int d0, d1;
        probe_cvt8(fa, s_a, d0, d1);
        av[2 * q] = d0; av[2 * q + 1] = d1;
        probe_cvt8(fb, s_b, d0, d1);
        bv[2 * q] = d0; bv[2 * q + 1] = d1;
    }
    asm volatile("" : "+v"(s_a), "+v"(s_b));
    v16f c;
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 0, 0, 0, s_a, 0, s_b);
#pragma unroll
    for (int i = 0; i < 16; ++i) c_out[lane * 16 + i] = c[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) { regs_out[lane * 16 + i] = av[i]; regs_out[lane * 16 + 8 + i] = bv[i]; }
}
}  // namespace

extern "C" int svps_probe_mx_block(const void* a, const void* b, const void* sa, const void* sb, float* c_out, int* regs_out, void* stream) {
    if (!a || !b || !sa || !sb || !c_out || !regs_out) return SVPS_ERR_BAD_ARG;
    hipLaunchKernelGGL(mx_block_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), static_cast<const _Float16*>(a),
                       static_cast<const _Float16*>(b), static_cast<const int*>(sa), static_cast<const int*>(sb), c_out, regs_out);
    return (int)hipGetLastError();
}

extern "C" int svps_probe_cvt_fp8(const void* x_pairs, float scale, int* out, int n, void* stream) {
    if (!x_pairs || !out) return SVPS_ERR_BAD_ARG;
    if (n <= 0) return SVPS_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(cvt_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const h2_t*>(x_pairs), scale, out, n);
    return (int)hipGetLastError();
}

extern "C" int svps_probe_mx_fp8(const void* a_regs, const void* b_regs, const void* scale_a, const void* scale_b, float* c_out, int n,
                                 void* stream) {
    if (!a_regs || !b_regs || !scale_a || !scale_b || !c_out) return SVPS_ERR_BAD_ARG;
    if (n <= 0) return SVPS_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(mx_probe_kernel, dim3(n), dim3(64), 0, static_cast<hipStream_t>(stream), static_cast<const int*>(a_regs),
                       static_cast<const int*>(b_regs), static_cast<const int*>(scale_a), static_cast<const int*>(scale_b), c_out);
    return (int)hipGetLastError();
}
