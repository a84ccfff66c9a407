// K8f - the feed-forward block of the slot update in ONE launch, for gfx950.
//
//     y = LN( pre + W2 act(W1 x + b1) + b2 ) * gamma + beta  (+ post)
//
// mmdet/models/detectors/dynamic_mask_head.py:379-385 (MaskRCNNHead: linear2(activation(linear1(x))) + residual + norm3) and
// :519-525 (TemporalSlotsHead, the same block with its own width). As two K8 launches (csrc/slot_gemm.hip) the hidden tensor
// [T * L, 2048] fp32 is written and read back (131 MB each way per stage at 16 000 rows) and every workgroup streams the whole
// weight of its layer; here a workgroup owns 64 rows and walks the hidden dimension in chunks of 256 columns:
//     GEMM1: h_c = act(x W1_c^T + b1_c)      64 x 256, K = 256, A = the x tile (bf16 hi / lo in LDS for the whole kernel)
//     GEMM2: acc += h_c W2_c^T               64 x 256, K = 256 of the hidden dimension, A = h_c as bf16 hi / lo in LDS
// with the arithmetic of K8 operation for operation (split-bf16: x_hi w_hi + x_lo w_hi + x_hi w_lo per k-step, k ascending, fp32
// accumulation, the same bias / activation / hi-lo split of the hidden value, the LayerNorm epilogue of svps_slot_gemm_ln), so
// the result is BITWISE the two-launch form (tests/test_row_ln_gpu.py).
//
// Mapping: 8 waves; wave w owns column block w (32 columns) of both products for BOTH 32-row blocks, so every B fragment of the
// two weights is requested by exactly one wave of the workgroup (fragment order, 1 KiB per wave instruction, register
// double-buffer one group of four k-steps ahead) and the A fragments come from LDS (padded 528-byte rows: conflict-free 16-byte
// reads). Software pipeline over the chunks: GEMM2(c) and GEMM1(c + 2) run back to back (192 MFMAs per wave without a barrier),
// the activation and the hi / lo split of h_{c+1} (raw block held in registers since the previous iteration) sit between the
// four groups of GEMM2(c), and only the LDS writes of h_{c+1} are left between the two barriers of a chunk.
#include <type_traits>
#include "common.h"
#include "../../include/slotvps_hip.h"

namespace svps {

constexpr int kFfRows = 64;
constexpr int kFfRow = 256 * 2 + 16;                 // bytes per staged row (256 bf16 + pad)
struct FfnLds {
    static constexpr int half = kFfRows * kFfRow;    // one of (hi, lo)
    static constexpr int xt = 0;                     // x tile: hi | lo
    static constexpr int ht = 2 * half;              // hidden chunk: hi | lo (reused by the LN epilogue as the fp32 out tile)
    static constexpr int total = 4 * half;           // 135 168 B
    static constexpr int ln_row = 256 * 4 + 16;
};
static_assert(kFfRows * FfnLds::ln_row <= 2 * FfnLds::half, "the LN epilogue reuses the hidden tile");

struct FfnArgs {
    const float* x;        // [M, 256]
    const __bf16* w1p;     // pack_b_fragments(W1 [H, 256])
    const float* b1;       // [H] or null
    const __bf16* w2p;     // pack_b_fragments(W2 [256, H])
    const float* b2;       // [256] or null
    const float* pre;      // [M, 256] added before the LayerNorm, or null
    const float* post;     // [M, 256] added after it, or null
    const float* gamma;
    const float* beta;
    float* y;              // [M, 256]
    float eps;
    int M, H;
};

__device__ __forceinline__ float ffn_wave_sum(float x) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m);
    return x;
}


// ST: the 16-bit type both operands are split into: __bf16 (hi + lo: 16 bits of mantissa) or _Float16 (22 bits; |x| < 65 504: precision
// "fp16x2", round 4) - three MFMAs per product either way
template <int ACT, typename ST = __bf16>   // ACT: 1 ReLU, 2 GELU (erf)
__global__ __launch_bounds__(512) void slot_ffn_kernel(FfnArgs a) {
    typedef ST stx8 __attribute__((ext_vector_type(8)));
    typedef ST stx4 __attribute__((ext_vector_type(4)));
    auto mfma_st = [](stx8 x_, stx8 w_, f32x16 c_) {
        if constexpr (__is_same(ST, _Float16)) return __builtin_amdgcn_mfma_f32_32x32x16_f16(x_, w_, c_, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x_, w_, c_, 0, 0, 0);
    };
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * kFfRows;
    const int NC = a.H >> 8;                                     // hidden chunks of 256 columns
    const int KS2 = a.H >> 4;                                    // k-steps of the second product

    const u32x4* w1s = reinterpret_cast<const u32x4*>(a.w1p) + lane;
    const u32x4* w2s = reinterpret_cast<const u32x4*>(a.w2p) + lane;
    // one group = four k-steps of one column block: 8 fragments (hi, lo per k-step); f = fragment index (cb * KS + ks) of the first
    auto load_w = [&](const u32x4* base, size_t f, u32x4 (&wb)[8]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            wb[2 * u] = base[(f + u) * 128];
            wb[2 * u + 1] = base[(f + u) * 128 + 64];
        }
    };
    // four k-steps (ks0 .. ks0 + 3 of the staged tile at `tile`) for both row blocks, K8's order of products per accumulator
    auto mma4 = [&](const char* tile, int ks0, const u32x4 (&wb)[8], f32x16 (&acc)[2]) {
        const char* ah = tile + r * kFfRow + 32 * ks0 + 16 * h;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const stx8 wh = __builtin_bit_cast(stx8, wb[2 * u]);
            const stx8 wl = __builtin_bit_cast(stx8, wb[2 * u + 1]);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const stx8 xh = *reinterpret_cast<const stx8*>(ah + 32 * b * kFfRow + 32 * u);
                const stx8 xl = *reinterpret_cast<const stx8*>(ah + 32 * b * kFfRow + 32 * u + FfnLds::half);
                acc[b] = mfma_st(xh, wh, acc[b]);
                acc[b] = mfma_st(xl, wh, acc[b]);
                acc[b] = mfma_st(xh, wl, acc[b]);
            }
        }
    };

    u32x4 wA[8], wB[8];
    load_w(w1s, (size_t)w * 16, wA);                             // GEMM1(0), group 0 - requested before the x tile

    // ---- x tile -> bf16 hi / lo in LDS (the split of K8's staging)
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
            stx4 vh, vl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                vh[e] = (ST)av[i][e];
                vl[e] = (ST)(av[i][e] - (float)vh[e]);
            }
            *reinterpret_cast<stx4*>(smem + FfnLds::xt + row * kFfRow + kg * 2) = vh;
            *reinterpret_cast<stx4*>(smem + FfnLds::xt + FfnLds::half + row * kFfRow + kg * 2) = vl;
        }
    }
    __syncthreads();

    f32x16 acc1[2], acc2[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc1[b][i] = 0.f; acc2[b][i] = 0.f; }

    uint32_t hp[2][8], lp[2][8];                                 // the activated hidden values of this wave's column, split, two per register
    // a quarter (q = 0 .. 3) of: bias + activation + hi / lo split of the raw GEMM1 block in acc1 (chunk c); clears what it read
    auto activate = [&](float bv, int q) {
        const int b = q >> 1, i0 = 8 * (q & 1);
#pragma unroll
        for (int i = i0; i < i0 + 8; i += 2) {
            float v[2];
            uint32_t hb[2], lb[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                v[e] = acc1[b][i + e] + bv;
                if constexpr (ACT == 1) v[e] = v[e] > 0.f ? v[e] : 0.f;
                if constexpr (ACT == 2) v[e] = svps_gelu_erf(v[e]);
                const ST hi = (ST)v[e];
                const ST lo = (ST)(v[e] - (float)hi);
                hb[e] = __builtin_bit_cast(unsigned short, hi);
                lb[e] = __builtin_bit_cast(unsigned short, lo);
                acc1[b][i + e] = 0.f;
            }
            hp[b][i >> 1] = hb[0] | (hb[1] << 16);
            lp[b][i >> 1] = lb[0] | (lb[1] << 16);
        }
    };
    auto bias1 = [&](int c) { return a.b1 ? a.b1[256 * c + 32 * w + r] : 0.f; };
    auto write_hidden = [&]() {
        char* base = smem + FfnLds::ht + (32 * w + r) * 2;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = 32 * b + (i & 3) + 8 * (i >> 2) + 4 * h;
                const uint32_t hw_ = hp[b][i >> 1], lw_ = lp[b][i >> 1];
                *reinterpret_cast<unsigned short*>(base + row * kFfRow) = (unsigned short)((i & 1) ? hw_ >> 16 : hw_);
                *reinterpret_cast<unsigned short*>(base + row * kFfRow + FfnLds::half) = (unsigned short)((i & 1) ? lw_ >> 16 : lw_);
            }
    };
    // GEMM1(c) for this wave's hidden column block 8 c + w; on entry wA holds its group 0; on exit wA holds `next`'s group 0
    auto gemm1 = [&](int c, const u32x4* nbase, size_t nf) {
        const size_t f = (size_t)(8 * c + w) * 16;
        load_w(w1s, f + 4, wB);
        mma4(smem + FfnLds::xt, 0, wA, acc1);
        load_w(w1s, f + 8, wA);
        mma4(smem + FfnLds::xt, 4, wB, acc1);
        load_w(w1s, f + 12, wB);
        mma4(smem + FfnLds::xt, 8, wA, acc1);
        load_w(nbase, nf, wA);
        mma4(smem + FfnLds::xt, 12, wB, acc1);
    };

    // ---- prologue: h_0 into LDS, the raw GEMM1 block of chunk 1 into acc1
    gemm1(0, NC > 1 ? w1s : w2s, NC > 1 ? (size_t)(8 + w) * 16 : (size_t)w * KS2);
    {
        const float bv = bias1(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) activate(bv, q);
    }
    write_hidden();
    if (NC > 1) gemm1(1, w2s, (size_t)w * KS2);                  // next: GEMM2(0), group 0
    __syncthreads();

    // ---- chunk c: GEMM2(c) from h_c in LDS, with the activation of the raw block of chunk c + 1 (in acc1) placed between its
    // four groups (vector work in the shadow of the matrix work); then GEMM1(c + 2) into the cleared acc1; between the two
    // barriers only the LDS writes of h_{c+1}
    // (the last two chunks have no GEMM1 / no activation left: the body is instantiated three times, branch-free inside, so that
    // the compiler schedules the vector work between the MFMAs)
    auto chunk = [&](int c, auto more_t, auto more2_t) {
        constexpr bool more = decltype(more_t)::value, more2 = decltype(more2_t)::value;
        const size_t f2 = (size_t)w * KS2 + 16 * c;              // GEMM2(c): column block w, k-steps 16 c ..
        float bv = 0.f;
        if constexpr (more) bv = bias1(c + 1);
        load_w(w2s, f2 + 4, wB);
        mma4(smem + FfnLds::ht, 0, wA, acc2);
        if constexpr (more) activate(bv, 0);
        load_w(w2s, f2 + 8, wA);
        mma4(smem + FfnLds::ht, 4, wB, acc2);
        if constexpr (more) activate(bv, 1);
        load_w(w2s, f2 + 12, wB);
        mma4(smem + FfnLds::ht, 8, wA, acc2);
        if constexpr (more) activate(bv, 2);
        if constexpr (more2) load_w(w1s, (size_t)(8 * (c + 2) + w) * 16, wA);
        else if constexpr (more) load_w(w2s, f2 + 16, wA);
        mma4(smem + FfnLds::ht, 12, wB, acc2);
        if constexpr (more) activate(bv, 3);
        if constexpr (more2) gemm1(c + 2, w2s, f2 + 16);         // next: GEMM2(c + 1), group 0
        __syncthreads();                                         // every wave is done reading h_c
        if constexpr (more) write_hidden();
        __syncthreads();
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
#pragma unroll 1
    for (int c = 0; c + 2 < NC; ++c) chunk(c, T_{}, T_{});
    if (NC >= 2) chunk(NC - 2, T_{}, F_{});
    chunk(NC - 1, F_{}, F_{});

    // ---- LN epilogue: the arithmetic of svps_slot_gemm_ln (tile -> LDS, one wave per row)
    constexpr int RPW = kFfRows / 8;
    float4 pv[RPW], qv[RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int m = m0 + w + 8 * j;
        const size_t base = (size_t)(m < a.M ? m : a.M - 1) * 256 + 4 * lane;
        pv[j] = a.pre ? *reinterpret_cast<const float4*>(a.pre + base) : make_float4(0.f, 0.f, 0.f, 0.f);
        qv[j] = a.post ? *reinterpret_cast<const float4*>(a.post + base) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float4 ww = *reinterpret_cast<const float4*>(a.gamma + 4 * lane);
    const float4 bb = *reinterpret_cast<const float4*>(a.beta + 4 * lane);
    char* ot = smem + FfnLds::ht;
    {
        const int col = 32 * w + r;
        const float bv = a.b2 ? a.b2[col] : 0.f;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = 32 * b + (i & 3) + 8 * (i >> 2) + 4 * h;
                *reinterpret_cast<float*>(ot + row * FfnLds::ln_row + col * 4) = acc2[b][i] + bv;
            }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int row = w + 8 * j, m = m0 + row;
        if (m >= a.M) break;
        float4 v = *reinterpret_cast<const float4*>(ot + row * FfnLds::ln_row + 16 * lane);
        if (a.pre) { v.x += pv[j].x; v.y += pv[j].y; v.z += pv[j].z; v.w += pv[j].w; }
        const float mean = ffn_wave_sum(v.x + v.y + v.z + v.w) * (1.f / 256.f);
        const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
        const float var = ffn_wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.f / 256.f);
        const float rstd = rsqrtf(var + a.eps);
        float4 o = make_float4(dx * rstd * ww.x + bb.x, dy * rstd * ww.y + bb.y, dz * rstd * ww.z + bb.z, dw * rstd * ww.w + bb.w);
        if (a.post) { o.x += qv[j].x; o.y += qv[j].y; o.z += qv[j].z; o.w += qv[j].w; }
        *reinterpret_cast<float4*>(a.y + (size_t)m * 256 + 4 * lane) = o;
    }
}

}  // namespace svps

namespace {
template <int ACT, typename ST>
int launch_ffn(const svps::FfnArgs& a, hipStream_t stream) {
    static SvpsLdsAttr attr;
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(svps::slot_ffn_kernel<ACT, ST>), svps::FfnLds::total); e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((svps::slot_ffn_kernel<ACT, ST>), dim3((a.M + svps::kFfRows - 1) / svps::kFfRows), dim3(512), svps::FfnLds::total, stream, a);
    return (int)hipGetLastError();
}
}  // namespace

// the same block with both weights packed as fp16 hi + lo (ops.pack_b_fragments(..., "fp16")) and the activations split likewise
extern "C" int svps_slot_ffn_f16(const float* x, const void* w1pack, const float* b1, const void* w2pack, const float* b2,
                                 const float* pre, const float* post, const float* gamma, const float* beta, float eps, int act,
                                 float* y, int M, int H, void* stream_) {
    if (!x || !w1pack || !w2pack || !gamma || !beta || !y) return SVPS_ERR_BAD_ARG;
    if (M <= 0 || H <= 0 || (H & 255) || (act != 1 && act != 2)) return SVPS_ERR_BAD_SHAPE;
    const svps::FfnArgs a{x, static_cast<const __bf16*>(w1pack), b1, static_cast<const __bf16*>(w2pack), b2, pre, post, gamma, beta,
                          y, eps, M, H};
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    return act == 1 ? launch_ffn<1, _Float16>(a, stream) : launch_ffn<2, _Float16>(a, stream);
}

extern "C" int svps_slot_ffn(const float* x, const void* w1pack, const float* b1, const void* w2pack, const float* b2,
                             const float* pre, const float* post, const float* gamma, const float* beta, float eps, int act,
                             float* y, int M, int H, void* stream_) {
    if (!x || !w1pack || !w2pack || !gamma || !beta || !y) return SVPS_ERR_BAD_ARG;
    if (M <= 0 || H <= 0 || (H & 255) || (act != 1 && act != 2)) return SVPS_ERR_BAD_SHAPE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const svps::FfnArgs a{x, static_cast<const __bf16*>(w1pack), b1, static_cast<const __bf16*>(w2pack), b2, pre, post, gamma, beta,
                          y, eps, M, H};
    const dim3 grid((M + svps::kFfRows - 1) / svps::kFfRows);
    if (act == 1) {
        static SvpsLdsAttr attr;
        if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(svps::slot_ffn_kernel<1>), svps::FfnLds::total); e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(svps::slot_ffn_kernel<1>, grid, dim3(512), svps::FfnLds::total, stream, a);
    } else {
        static SvpsLdsAttr attr;
        if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(svps::slot_ffn_kernel<2>), svps::FfnLds::total); e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(svps::slot_ffn_kernel<2>, grid, dim3(512), svps::FfnLds::total, stream, a);
    }
    return (int)hipGetLastError();
}
