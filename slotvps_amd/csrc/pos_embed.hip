// Sine position embedding (DETR style), reference: PositionEmbeddingSine.forward,
// mmdet/models/detectors/position_encoding.py:236-256 with normalize=True, scale=2*pi,
// temperature=1e4 and an all-False padding mask (vps_temporal_slots.py:141 builds the NestedTensor
// from an unpadded feature map). Pure function of (H, W): the host caches one map per level.
//
// Output is pixel-major [H*W, D] fp32 - the "b h w c" view the retriever adds to the features
// (dynamic_mask_head.py:430) - with channels [0, D/2) from the row index and [D/2, D) from the
// column index; even channels sin, odd channels cos; every fp32 operation in reference order.
#include <hip/hip_runtime.h>

#include "../../include/slotvps_hip.h"

namespace svps {

__global__ __launch_bounds__(256) void pos_embed_sine_kernel(float* __restrict__ out, int H, int W, int D) {
    const int half = D / 2;
    const size_t total = (size_t)H * W * D;
    const float two_pi = 6.283185307179586f;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % D);
        const int p = (int)(idx / D);
        const int y = p / W, x = p % W;
        const bool is_y = c < half;
        const int cc = is_y ? c : c - half;
        // cumsum of ones -> index + 1; divided by (last + eps), times scale
        const float num = (float)((is_y ? y : x) + 1);
        const float den = (float)(is_y ? H : W) + 1e-6f;
        const float e = num / den * two_pi;
        const float expo = 2.f * floorf((float)cc / 2.f) / (float)half;
        const float dim_t = powf(10000.f, expo);
        const float a = e / dim_t;
        out[idx] = (cc & 1) ? cosf(a) : sinf(a);
    }
}

// The embedding is separable: channels [0, D/2) depend on the row only, [D/2, D) on the column only.
// tab [n, D/2]: entry (i, c) = sin/cos(((i + 1) / (n + 1e-6) * 2 pi) / 10000^(2 floor(c/2) / (D/2)))
__global__ __launch_bounds__(256) void pos_embed_table_kernel(float* __restrict__ tab, int n, int half) {
    const int total = n * half;
    const float two_pi = 6.283185307179586f;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int cc = idx % half, i = idx / half;
        const float e = (float)(i + 1) / ((float)n + 1e-6f) * two_pi;
        const float dim_t = powf(10000.f, 2.f * floorf((float)cc / 2.f) / (float)half);
        const float a = e / dim_t;
        tab[idx] = (cc & 1) ? cosf(a) : sinf(a);
    }
}

}  // namespace svps

extern "C" int svps_pos_embed_sine_tables(float* ytab, float* xtab, int H, int W, int D, void* stream_) {
    if (!ytab || !xtab) return SVPS_ERR_BAD_ARG;
    if (H <= 0 || W <= 0 || D <= 0 || (D & 3)) return SVPS_ERR_BAD_SHAPE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int half = D / 2;
    hipLaunchKernelGGL(svps::pos_embed_table_kernel, dim3((H * half + 255) / 256), dim3(256), 0, stream, ytab, H, half);
    hipLaunchKernelGGL(svps::pos_embed_table_kernel, dim3((W * half + 255) / 256), dim3(256), 0, stream, xtab, W, half);
    return (int)hipGetLastError();
}

extern "C" int svps_pos_embed_sine(float* out, int H, int W, int D, void* stream_) {
    if (!out) return SVPS_ERR_BAD_ARG;
    if (H <= 0 || W <= 0 || D <= 0 || (D & 3)) return SVPS_ERR_BAD_SHAPE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const size_t total = (size_t)H * W * D;
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    svps_prof_mark(SVPS_KERNEL_POS_EMBED, 0, stream);
    hipLaunchKernelGGL(svps::pos_embed_sine_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, out, H, W, D);
    svps_prof_mark(SVPS_KERNEL_POS_EMBED, 1, stream);
    return (int)hipGetLastError();
}
