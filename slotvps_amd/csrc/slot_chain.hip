// K8c - chains of 256 -> 256 dense layers with LayerNorm on the slot side in ONE launch, for gfx950.
//
// The slot update runs short chains of   y = LN(x W^T + b [+ pre]) * gamma + beta  (+ReLU)  (+ post)   on the same [T * L, 256]
// rows (mmdet/models/detectors/dynamic_mask_head.py): the class / embedding towers (:394-397, four layers), the q / k / v
// projections of the temporal retriever (:555-557, three layers on one input), the output projection of the self-attention
// followed by the retriever's query projection (:356-358 then :431). As one K8 launch per layer (csrc/slot_gemm.hip,
// svps_slot_gemm_ln) each layer pays a launch, a read of its input rows from HBM and its share of the weight stream per 32-row
// workgroup. Here a workgroup owns 64 rows for the whole chain: the input tile and the running result stay in LDS as bf16 hi / lo
// operand tiles, every layer streams its weight once per workgroup, results leave for HBM only where a caller needs them.
//
// Arithmetic: svps_slot_gemm_ln operation for operation (split-bf16 products x_hi w_hi + x_lo w_hi + x_hi w_lo per k-step, k
// ascending, fp32 accumulation, bias, the two-pass LayerNorm of svps_row_ln, the hi / lo split of a layer's fp32 result as the
// next layer's operand) - BITWISE the per-layer launches (tests/test_row_ln_gpu.py).
//
// Mapping: 8 waves; wave w owns column block w (32 columns) for both 32-row blocks (every B fragment requested by one wave,
// register double buffer one group of four k-steps ahead; the first group of the NEXT layer is requested before this layer's
// epilogue). LDS: X = the chain's input tile (kept), Y = the running result (also the fp32 tile of the LayerNorm epilogue).
#include "common.h"
#include "../../include/slotvps_hip.h"

namespace svps {

constexpr int kChRows = 64;
constexpr int kChRow = 256 * 2 + 16;                 // bytes per staged operand row (256 bf16 + pad)
constexpr int kChMaxLayers = 6;
struct ChainLds {
    static constexpr int half = kChRows * kChRow;    // one of (hi, lo)
    static constexpr int xt = 0;
    static constexpr int yt = 2 * half;
    static constexpr int total = 4 * half;           // 135 168 B
    static constexpr int ln_row = 256 * 4 + 16;
};
static_assert(kChRows * ChainLds::ln_row <= 2 * ChainLds::half, "the LN epilogue's fp32 tile lives in the Y region");

struct ChainLayer {
    const __bf16* wpack;   // pack_b_fragments(W [256, 256])
    const float* bias;     // [256] or null
    const float* gamma;    // LayerNorm weight [256]
    const float* beta;     // LayerNorm bias [256]
    const float* pre;      // [M, 256] added before the LayerNorm, or null
    const float* post;     // [M, 256] added after it (and after the ReLU), or null
    float* out;            // [M, 256], or null: the result only feeds the next layer
    float eps;
    int relu;
    int src;               // 0: the chain's input rows; 1: the previous layer's result
};
struct ChainArgs {
    const float* x;        // [M, 256]
    int M, n;
    ChainLayer layer[kChMaxLayers];
};

__device__ __forceinline__ float ch_wave_sum(float x) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m);
    return x;
}

__global__ __launch_bounds__(512) void slot_chain_kernel(ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * kChRows;

    auto load_w = [&](const u32x4* base, int ks0, u32x4 (&wb)[8]) {   // k-steps ks0 .. ks0 + 3 of column block w
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            wb[2 * u] = base[(size_t)(w * 16 + ks0 + u) * 128];
            wb[2 * u + 1] = base[(size_t)(w * 16 + ks0 + u) * 128 + 64];
        }
    };
    auto mma4 = [&](const char* tile, int ks0, const u32x4 (&wb)[8], f32x16 (&acc)[2]) {
        const char* ah = tile + r * kChRow + 32 * ks0 + 16 * h;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bf16x8 wh = __builtin_bit_cast(bf16x8, wb[2 * u]);
            const bf16x8 wl = __builtin_bit_cast(bf16x8, wb[2 * u + 1]);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const bf16x8 xh = *reinterpret_cast<const bf16x8*>(ah + 32 * b * kChRow + 32 * u);
                const bf16x8 xl = *reinterpret_cast<const bf16x8*>(ah + 32 * b * kChRow + 32 * u + ChainLds::half);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, wh, acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, wh, acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, wl, acc[b], 0, 0, 0);
            }
        }
    };

    u32x4 wA[8], wB[8];
    load_w(reinterpret_cast<const u32x4*>(a.layer[0].wpack) + lane, 0, wA);

    // ---- the chain's input tile -> bf16 hi / lo in LDS (the split of K8's staging)
    {
        f32x4 av[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = tid + 512 * i, row = q >> 6, kg = (q & 63) * 4;
            const int m = m0 + row;
            av[i] = m < a.M ? *reinterpret_cast<const f32x4*>(a.x + (size_t)m * 256 + kg) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = tid + 512 * i, row = q >> 6, kg = (q & 63) * 4;
            bf16x4 vh, vl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                vh[e] = (__bf16)av[i][e];
                vl[e] = (__bf16)(av[i][e] - (float)vh[e]);
            }
            *reinterpret_cast<bf16x4*>(smem + ChainLds::xt + row * kChRow + kg * 2) = vh;
            *reinterpret_cast<bf16x4*>(smem + ChainLds::xt + ChainLds::half + row * kChRow + kg * 2) = vl;
        }
    }
    __syncthreads();

    constexpr int RPW = kChRows / 8;                             // rows per wave in the epilogue
#pragma unroll 1
    for (int l = 0; l < a.n; ++l) {
        const ChainLayer L = a.layer[l];
        const u32x4* ws = reinterpret_cast<const u32x4*>(L.wpack) + lane;
        const char* tile = smem + (L.src ? ChainLds::yt : ChainLds::xt);
        f32x16 acc[2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
        load_w(ws, 4, wB);
        mma4(tile, 0, wA, acc);
        load_w(ws, 8, wA);
        mma4(tile, 4, wB, acc);
        load_w(ws, 12, wB);
        mma4(tile, 8, wA, acc);
        if (l + 1 < a.n) load_w(reinterpret_cast<const u32x4*>(a.layer[l + 1].wpack) + lane, 0, wA);
        mma4(tile, 12, wB, acc);

        // ---- LN epilogue (svps_slot_gemm_ln's): residual rows and the affine pair requested first
        float4 pv[RPW], qv[RPW];
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const int m = m0 + w + 8 * j;
            const size_t base = (size_t)(m < a.M ? m : a.M - 1) * 256 + 4 * lane;
            pv[j] = L.pre ? *reinterpret_cast<const float4*>(L.pre + base) : make_float4(0.f, 0.f, 0.f, 0.f);
            qv[j] = L.post ? *reinterpret_cast<const float4*>(L.post + base) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float4 ww = *reinterpret_cast<const float4*>(L.gamma + 4 * lane);
        const float4 bb = *reinterpret_cast<const float4*>(L.beta + 4 * lane);
        __syncthreads();                                         // every wave is done with the operand tile (Y may be the source)
        char* ot = smem + ChainLds::yt;
        {
            const int col = 32 * w + r;
            const float bv = L.bias ? L.bias[col] : 0.f;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = 32 * b + (i & 3) + 8 * (i >> 2) + 4 * h;
                    *reinterpret_cast<float*>(ot + row * ChainLds::ln_row + col * 4) = acc[b][i] + bv;
                }
        }
        __syncthreads();
        float4 res[RPW];
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const int row = w + 8 * j, m = m0 + row;
            float4 v = *reinterpret_cast<const float4*>(ot + row * ChainLds::ln_row + 16 * lane);
            if (L.pre) { v.x += pv[j].x; v.y += pv[j].y; v.z += pv[j].z; v.w += pv[j].w; }
            const float mean = ch_wave_sum(v.x + v.y + v.z + v.w) * (1.f / 256.f);
            const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
            const float var = ch_wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.f / 256.f);
            const float rstd = rsqrtf(var + L.eps);
            float4 o = make_float4(dx * rstd * ww.x + bb.x, dy * rstd * ww.y + bb.y, dz * rstd * ww.z + bb.z, dw * rstd * ww.w + bb.w);
            if (L.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
            if (L.post) { o.x += qv[j].x; o.y += qv[j].y; o.z += qv[j].z; o.w += qv[j].w; }
            if (L.out && m < a.M) *reinterpret_cast<float4*>(L.out + (size_t)m * 256 + 4 * lane) = o;
            res[j] = o;
        }
        if (l + 1 == a.n) break;
        __syncthreads();                                         // every row of the fp32 tile has been read
#pragma unroll
        for (int j = 0; j < RPW; ++j) {                          // the result as the next layer's operand rows (rows >= M: harmless values, never stored)
            const int row = w + 8 * j;
            const float e4[4] = {res[j].x, res[j].y, res[j].z, res[j].w};
            bf16x4 vh, vl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                vh[e] = (__bf16)e4[e];
                vl[e] = (__bf16)(e4[e] - (float)vh[e]);
            }
            *reinterpret_cast<bf16x4*>(smem + ChainLds::yt + row * kChRow + 8 * lane) = vh;
            *reinterpret_cast<bf16x4*>(smem + ChainLds::yt + ChainLds::half + row * kChRow + 8 * lane) = vl;
        }
        __syncthreads();
    }
}

}  // namespace svps

extern "C" int svps_slot_chain(const float* x, int M, int n_layers, const void* const* wpack, const float* const* bias,
                               const float* const* gamma, const float* const* beta, const float* eps, const int* relu,
                               const float* const* pre, const float* const* post, float* const* out, const int* src, void* stream_) {
    if (!x || !wpack || !bias || !gamma || !beta || !eps || !relu || !pre || !post || !out || !src) return SVPS_ERR_BAD_ARG;
    if (M <= 0 || n_layers <= 0 || n_layers > svps::kChMaxLayers) return SVPS_ERR_BAD_SHAPE;
    svps::ChainArgs a{};
    a.x = x;
    a.M = M;
    a.n = n_layers;
    for (int l = 0; l < n_layers; ++l) {
        if (!wpack[l] || !gamma[l] || !beta[l]) return SVPS_ERR_BAD_ARG;
        if ((src[l] != 0 && src[l] != 1) || (l == 0 && src[l] != 0)) return SVPS_ERR_BAD_SHAPE;
        if (l + 1 == n_layers && !out[l]) return SVPS_ERR_BAD_ARG;           // the last result must go somewhere
        a.layer[l] = svps::ChainLayer{static_cast<const __bf16*>(wpack[l]), bias[l], gamma[l], beta[l], pre[l], post[l], out[l], eps[l],
                                      relu[l], src[l]};
    }
    // a layer with src = 1 reads the result of the layer before it, whatever that layer's own source was
    static SvpsLdsAttr attr;
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(svps::slot_chain_kernel), svps::ChainLds::total); e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(svps::slot_chain_kernel, dim3((M + svps::kChRows - 1) / svps::kChRows), dim3(512), svps::ChainLds::total,
                       static_cast<hipStream_t>(stream_), a);
    return (int)hipGetLastError();
}
