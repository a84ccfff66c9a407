// K5 - fused row operators of the slot side (rows of 256 fp32 values: slots x channels).
//
// The reference's slot update (mmdet/models/detectors/dynamic_mask_head.py:342-400, :494-527, :550-572)
// is a chain of  residual add -> LayerNorm -> (ReLU) -> (cast)  steps between small GEMMs; as separate
// framework kernels each costs a launch (~4.5 us) for ~0.5 MB of data. svps_row_ln fuses one such step:
//
//     y = LN(x [+ pre])  * w[g] + b[g]      LayerNorm over the 256 channels, biased variance, two-pass, eps
//     y = relu(y)                            (flag)
//     y = y + post                           (optional second residual, e.g. S + U' of :317)
//     out_f32 / out_bf16 = y
//
// with g = row / rows_per_group selecting one of G affine pairs (batched launches over e.g. the q/k/v
// projections of the temporal retriever, or the first layers of the class and embedding towers).
// One wavefront per row (64 lanes x float4), reductions with cross-lane shuffles only.
#include <hip/hip_runtime.h>

#include "../../include/slotvps_hip.h"

namespace svps {

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m);
    return x;
}

__global__ __launch_bounds__(256) void row_ln_kernel(const float* __restrict__ x, const float* __restrict__ pre,
                                                     const float* __restrict__ post, const float* __restrict__ w,
                                                     const float* __restrict__ b, float eps, int relu,
                                                     int rows, int rows_per_group, float* __restrict__ out_f32,
                                                     __bf16* __restrict__ out_bf16) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const size_t base = (size_t)row * 256 + 4 * lane;
    float4 v = *reinterpret_cast<const float4*>(x + base);
    if (pre) {
        const float4 p = *reinterpret_cast<const float4*>(pre + base);
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.f / 256.f);
    const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
    const float var = wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.f / 256.f);
    const float rstd = rsqrtf(var + eps);
    const int g = row / rows_per_group;
    const float4 ww = *reinterpret_cast<const float4*>(w + (size_t)g * 256 + 4 * lane);
    const float4 bb = *reinterpret_cast<const float4*>(b + (size_t)g * 256 + 4 * lane);
    float4 y = make_float4(dx * rstd * ww.x + bb.x, dy * rstd * ww.y + bb.y, dz * rstd * ww.z + bb.z,
                           dw * rstd * ww.w + bb.w);
    if (relu) {
        y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f);
    }
    if (post) {
        const float4 p = *reinterpret_cast<const float4*>(post + base);
        y.x += p.x; y.y += p.y; y.z += p.z; y.w += p.w;
    }
    if (out_f32) *reinterpret_cast<float4*>(out_f32 + base) = y;
    if (out_bf16) {
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
        bf16x4_t o;
        o[0] = (__bf16)y.x; o[1] = (__bf16)y.y; o[2] = (__bf16)y.z; o[3] = (__bf16)y.w;
        *reinterpret_cast<bf16x4_t*>(out_bf16 + base) = o;
    }
}

}  // namespace svps

extern "C" int svps_row_ln(const float* x, const float* pre, const float* post, const float* w, const float* b,
                           float eps, int relu, int rows, int rows_per_group, int D, float* out_f32, void* out_bf16,
                           void* stream_) {
    if (!x || !w || !b || (!out_f32 && !out_bf16)) return SVPS_ERR_BAD_ARG;
    if (D != 256 || rows <= 0 || rows_per_group <= 0) return SVPS_ERR_BAD_SHAPE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(svps::row_ln_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, x, pre, post, w, b, eps, relu,
                       rows, rows_per_group, out_f32, static_cast<__bf16*>(out_bf16));
    return (int)hipGetLastError();
}
