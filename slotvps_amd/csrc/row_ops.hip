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

// ---- softmax over the last index of [rows, cols] fp32 (the temporal retriever's softmax over the query axis,
// dynamic_mask_head.py:559-567, applied to the transposed logits): one wavefront per row, torch.softmax's arithmetic
// (max, exp(x - max), sum), the division as a multiplication by the reciprocal of the sum (<= 1 ulp from torch's divide). y may BE x
// (ops.row_softmax(inplace=True)): the pointers are not declared __restrict__, and a lane only ever re-reads the element it writes.
// Rows are 2 KB: the second and third read come from L1 / L2.
// `scale` (a power of two): y = scale * softmax(x) - for a consumer that carries the probabilities as fp16 hi + lo (svps_bgemm_f16): below
// 6.1e-5 fp16 is subnormal (absolute resolution 6e-8), and a query that receives almost no mass from any key has its whole output row
// in that range while the LayerNorm behind it scales the row back up by up to 1 / sqrt(eps).
__global__ __launch_bounds__(256) void row_softmax_kernel(const float* x, float* y, int rows, int cols, float scale) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * cols;
    float* yr = y + (size_t)row * cols;
    float m = -INFINITY;
    for (int c = lane; c < cols; c += 64) m = fmaxf(m, xr[c]);
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) m = fmaxf(m, __shfl_xor(m, k));
    float sum = 0.f;
    for (int c = lane; c < cols; c += 64) sum += expf(xr[c] - m);
    sum = wave_sum(sum);
    const float inv = (1.f / sum) * scale;                   // (scale = 1: unchanged; a power of two: exact)
    for (int c = lane; c < cols; c += 64) yr[c] = expf(xr[c] - m) * inv;
}

// ---- query side of the statistics-fused retriever (retr_attn.hip) -------------------------------------------------
// One wavefront per (frame, padded slot row). From x = to_q(slots) (the GEMM stays a library call):
//     q = norm_q(x)                         MaskDynamicConv.forward :431
//     g = q * gamma_k   -> gp [T, LP, 256]  operand of the key fold  Q'' = g W~_k   (rows >= L written as zeros)
//     c3 = log2(e) q . beta_k (rows >= L: -1e30) -> [T, LP]          a1 = g . b~_k -> [T, LP]
__global__ __launch_bounds__(256) void retr_query_prep_kernel(const float* __restrict__ x,      // [T, L, 256]
                                                              const float* __restrict__ qw, const float* __restrict__ qb, float eps,
                                                              const float* __restrict__ gk, const float* __restrict__ bek,
                                                              const float* __restrict__ bck,
                                                              float* __restrict__ gp, float* __restrict__ c3, float* __restrict__ a1,
                                                              int T, int L, int LP) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T * LP) return;
    const int t = row / LP, l = row - t * LP;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    float c3v = 0.f, a1v = 0.f;
    if (l < L) {
        const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)t * L + l) * 256 + 4 * lane);
        const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.f / 256.f);
        const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
        const float rstd = rsqrtf(wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.f / 256.f) + eps);
        const float4 w4 = *reinterpret_cast<const float4*>(qw + 4 * lane), b4 = *reinterpret_cast<const float4*>(qb + 4 * lane);
        const float4 q = make_float4(dx * rstd * w4.x + b4.x, dy * rstd * w4.y + b4.y, dz * rstd * w4.z + b4.z, dw * rstd * w4.w + b4.w);
        const float4 k4 = *reinterpret_cast<const float4*>(gk + 4 * lane), e4 = *reinterpret_cast<const float4*>(bek + 4 * lane);
        const float4 c4 = *reinterpret_cast<const float4*>(bck + 4 * lane);
        g = make_float4(q.x * k4.x, q.y * k4.y, q.z * k4.z, q.w * k4.w);
        c3v = wave_sum(q.x * e4.x + q.y * e4.y + q.z * e4.z + q.w * e4.w);
        a1v = wave_sum(g.x * c4.x + g.y * c4.y + g.z * c4.z + g.w * c4.w);
    }
    *reinterpret_cast<float4*>(gp + (size_t)row * 256 + 4 * lane) = g;
    // c3 leaves in the form K1' consumes: log2(e) * q . beta_k (its softmax runs on exp2), and -1e30 in the padded rows, which
    // is what removes them from every pixel's softmax without a mask in the kernel
    if (lane == 0) { c3[row] = l < L ? c3v * 1.4426950408889634f : -1.0e30f; a1[row] = a1v; }
}

// Q'' [rows, 256] fp32 -> FP16 hi and lo = fp16(Q'' - hi), the two A operands of K1' (22 bits of mantissa together)
__global__ __launch_bounds__(256) void retr_split_kernel(const float* __restrict__ q2, _Float16* __restrict__ hi, _Float16* __restrict__ lo, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
    const float4 v = reinterpret_cast<const float4*>(q2)[i];
    f16x4_t h, l;
    h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
    l[0] = (_Float16)(v.x - (float)h[0]); l[1] = (_Float16)(v.y - (float)h[1]); l[2] = (_Float16)(v.z - (float)h[2]); l[3] = (_Float16)(v.w - (float)h[3]);
    reinterpret_cast<f16x4_t*>(hi)[i] = h;
    reinterpret_cast<f16x4_t*>(lo)[i] = l;
}

// ---- slot self-attention (nn.MultiheadAttention of a stage, dynamic_mask_head.py:346-355) -------------------------------
// softmax(q k^T / sqrt(hd)) v for one (frame, head) per workgroup on the packed projection qkv [T, L, 3, nh, hd] (hd = 32),
// on the matrix cores with fp32-class precision: every operand is carried as bf16 hi + lo and the three significant
// products are accumulated in fp32 (split-bf16, relative error ~1e-5 of the largest term). NW waves, wave w owns the 32
// queries [32w, 32w + 32); L <= 32 NW slots.
//   1. S^T = K Q'^T   (Q' = q * log2(e) / sqrt(hd)): keys are MFMA ROWS, queries MFMA COLUMNS = lanes, so the softmax over
//      keys of a query is an in-lane reduction over the NW accumulator blocks + one lane^32 exchange (the layout K1 uses)
//   2. P = exp2(S^T - max), den = sum; P moves from the accumulator layout to the B-operand layout with v_permlane32_swap
//      (no LDS round trip): lane (query, h) ends up with the 8 consecutive keys 16 ks + 8 h .. of its k-step
//   3. O^T = V^T P    (V^T staged once per workgroup in LDS as bf16 hi / lo, dims x keys), O = O^T / den
// Replaces the framework's generic attention kernel for these tiny shapes (100 x 100 x 32 per head). Measured for 80 frames
// x 8 heads (tools/self_attn_probe.py): 15 us, against 30 us for the framework's fused attention kernel, 112 us for
// matmul + softmax + matmul through the GEMM library and 81 us for a one-thread-per-query fp32 vector-ALU kernel (the first
// version of this function); 2.2e-5 from the fp32 result on O(1) outputs.
// ST = the 16-bit type of the operand split: __bf16 (hi + lo = 16 bits of mantissa; the 16-bit modes) or _Float16 (22 bits: the slot side of
// mode "fp16x2", round 5 - q / sqrt(32), k, v are O(1 ... 10) and the probabilities <= 1: inside fp16's range; a probability below
// fp16's normal range keeps an ABSOLUTE resolution of 6e-8, and nothing behind this kernel rescales a row)
typedef __attribute__((ext_vector_type(16))) float ra_f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 ra_b8;
typedef __attribute__((ext_vector_type(8))) _Float16 ra_h8;
__device__ __forceinline__ ra_f32x16 mfma16(ra_b8 a, ra_b8 b, ra_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ ra_f32x16 mfma16(ra_h8 a, ra_h8 b, ra_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

template <typename ST>
__device__ __forceinline__ void split8(const float* v, __attribute__((ext_vector_type(8))) ST& hi, __attribute__((ext_vector_type(8))) ST& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float x = v[j];
        asm volatile("" : "+v"(x));              // ONE fp32 value for both halves (see retr_attn.hip, p2_store)
        hi[j] = (ST)x;
        lo[j] = (ST)(x - (float)hi[j]);
    }
}

template <int NW, typename ST = __bf16>
__global__ __launch_bounds__(NW * 64) void slot_self_attn_kernel(const float* __restrict__ qkv, float* __restrict__ out, int L, int nh, float scale_log2e) {
    typedef __attribute__((ext_vector_type(8))) ST ra_bf16x8;        // (8 x the split type)
    constexpr int HD = 32, LKP = 32 * NW;                             // padded key count
    constexpr int kVtRow = LKP * 2 + 16;                              // bytes per dim row of V^T (padded: conflict-free 16-byte reads)
    __shared__ __attribute__((aligned(16))) char vt[2 * HD * kVtRow]; // [hi | lo][dim][key] bf16
    const int t = blockIdx.y, hh = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
    const size_t rs = (size_t)3 * nh * HD;                            // floats per (frame, slot) row of qkv
    const float* base = qkv + (size_t)t * L * rs + hh * HD;
    // ---- V^T -> LDS (hi / lo), keys past L zero
    for (int i = tid; i < LKP * (HD / 4); i += NW * 64) {
        const int key = i / (HD / 4), d4 = i - key * (HD / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (key < L) v = *reinterpret_cast<const float4*>(base + (size_t)key * rs + (size_t)2 * nh * HD + 4 * d4);
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float xv = vv[e];
            asm volatile("" : "+v"(xv));
            const ST hi = (ST)xv;
            *reinterpret_cast<ST*>(vt + (4 * d4 + e) * kVtRow + key * 2) = hi;
            *reinterpret_cast<ST*>(vt + HD * kVtRow + (4 * d4 + e) * kVtRow + key * 2) = (ST)(xv - (float)hi);
        }
    }
    // ---- this wave's queries as B fragments (hi / lo), pre-scaled
    ra_bf16x8 qh[2], ql[2];
    {
        const int qi = 32 * w + r;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (qi < L) {
                const float4 a = *reinterpret_cast<const float4*>(base + (size_t)qi * rs + 16 * ks + 8 * h);
                const float4 b = *reinterpret_cast<const float4*>(base + (size_t)qi * rs + 16 * ks + 8 * h + 4);
                v[0] = a.x * scale_log2e; v[1] = a.y * scale_log2e; v[2] = a.z * scale_log2e; v[3] = a.w * scale_log2e;
                v[4] = b.x * scale_log2e; v[5] = b.y * scale_log2e; v[6] = b.z * scale_log2e; v[7] = b.w * scale_log2e;
            }
            split8<ST>(v, qh[ks], ql[ks]);
        }
    }
    // ---- S^T blocks: keys x queries
    ra_f32x16 acc[NW];
#pragma unroll
    for (int nb = 0; nb < NW; ++nb) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
        const int key = 32 * nb + r;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (key < L) {
                const float4 a = *reinterpret_cast<const float4*>(base + (size_t)key * rs + (size_t)nh * HD + 16 * ks + 8 * h);
                const float4 b = *reinterpret_cast<const float4*>(base + (size_t)key * rs + (size_t)nh * HD + 16 * ks + 8 * h + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            }
            ra_bf16x8 kh, kl;
            split8<ST>(v, kh, kl);
            acc[nb] = mfma16(kh, qh[ks], acc[nb]);
            acc[nb] = mfma16(kl, qh[ks], acc[nb]);
            acc[nb] = mfma16(kh, ql[ks], acc[nb]);
        }
    }
    // ---- softmax over keys (rows), per query (lane column): rows of register i: 32 nb + (i & 3) + 8 (i >> 2) + 4 h
    float m = -1.0e30f;
#pragma unroll
    for (int nb = 0; nb < NW; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bool ok = 32 * nb + (i & 3) + 8 * (i >> 2) + 4 * h < L;
            acc[nb][i] = ok ? acc[nb][i] : -1.0e30f;
            m = fmaxf(m, acc[nb][i]);
        }
    {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
        m = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    float den = 0.f;
#pragma unroll
    for (int nb = 0; nb < NW; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[nb][i] = __builtin_amdgcn_exp2f(acc[nb][i] - m);      // masked rows: exp2(-1e30 - m) = 0
            den += acc[nb][i];
        }
    {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(den), __float_as_uint(den), false, false);
        den = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    __syncthreads();                                                  // V^T is in LDS
    // ---- O^T = V^T P: k-step (nb, ks2) covers keys 32 nb + 16 ks2 .. + 16
    ra_f32x16 o;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = 0.f;
#pragma unroll
    for (int nb = 0; nb < NW; ++nb)
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
            // accumulator layout -> B-operand layout: registers 8 ks2 + e (rows 16 ks2 + 4 h + e) and 8 ks2 + 4 + e (rows + 8);
            // v_permlane32_swap(X, Y): X' = {lower lanes: X, upper lanes: Y of the lower lanes}, Y' = {X of the upper lanes, Y}
            float p[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[nb][8 * ks2 + e]), __float_as_uint(acc[nb][8 * ks2 + 4 + e]),
                                                           false, false);
                p[e] = __uint_as_float(sw[0]);
                p[4 + e] = __uint_as_float(sw[1]);
            }
            ra_bf16x8 ph, pl;
            split8<ST>(p, ph, pl);
            const char* va = vt + r * kVtRow + (32 * nb + 16 * ks2 + 8 * h) * 2;
            const ra_bf16x8 vh = *reinterpret_cast<const ra_bf16x8*>(va);
            const ra_bf16x8 vl = *reinterpret_cast<const ra_bf16x8*>(va + HD * kVtRow);
            o = mfma16(vh, ph, o);
            o = mfma16(vl, ph, o);
            o = mfma16(vh, pl, o);
        }
    // ---- O[query][dim] = O^T / den: lane (query r, h) holds dims 8 g + 4 h + e
    const int qi = 32 * w + r;
    if (qi < L) {
        const float inv = 1.f / den;
        float* dst = out + ((size_t)t * L + qi) * nh * HD + hh * HD;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(dst + 8 * g + 4 * h) = make_float4(o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv);
    }
}

}  // namespace svps

extern "C" int svps_retr_query_prep(const float* x, const float* lnq_w, const float* lnq_b, float lnq_eps, const float* lnk_w,
                                    const float* lnk_b, const float* bck, float* gp, float* c3, float* a1, int T, int L, int LP,
                                    int D, void* stream_) {
    if (!x || !lnq_w || !lnq_b || !lnk_w || !lnk_b || !bck || !gp || !c3 || !a1) return SVPS_ERR_BAD_ARG;
    if (D != 256 || T <= 0 || L <= 0 || L > LP || (LP != 128 && LP != 256)) return SVPS_ERR_BAD_SHAPE;
    const int rows = T * LP;
    hipLaunchKernelGGL(svps::retr_query_prep_kernel, dim3((rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream_), x, lnq_w,
                       lnq_b, lnq_eps, lnk_w, lnk_b, bck, gp, c3, a1, T, L, LP);
    return (int)hipGetLastError();
}

extern "C" int svps_retr_split(const float* q2, void* hi, void* lo, size_t n, void* stream_) {
    if (!q2 || !hi || !lo) return SVPS_ERR_BAD_ARG;
    if (n == 0 || (n & 3)) return SVPS_ERR_BAD_SHAPE;
    const size_t n4 = n / 4;
    hipLaunchKernelGGL(svps::retr_split_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream_), q2,
                       static_cast<_Float16*>(hi), static_cast<_Float16*>(lo), n4);
    return (int)hipGetLastError();
}

namespace {
template <typename ST>
int launch_self_attn(const float* qkv, float* out, int T, int L, int nheads, int head_dim, void* stream_) {
    if (!qkv || !out) return SVPS_ERR_BAD_ARG;
    if (T <= 0 || L <= 0 || L > 256 || nheads <= 0 || head_dim != 32) return SVPS_ERR_BAD_SHAPE;
    const float sl2 = 1.4426950408889634f / sqrtf((float)head_dim);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (L <= 128)
        hipLaunchKernelGGL((svps::slot_self_attn_kernel<4, ST>), dim3(nheads, T), dim3(256), 0, stream, qkv, out, L, nheads, sl2);
    else
        hipLaunchKernelGGL((svps::slot_self_attn_kernel<8, ST>), dim3(nheads, T), dim3(512), 0, stream, qkv, out, L, nheads, sl2);
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int svps_slot_self_attn(const float* qkv, float* out, int T, int L, int nheads, int head_dim, void* stream_) {
    return launch_self_attn<__bf16>(qkv, out, T, L, nheads, head_dim, stream_);
}
// the same with fp16 hi + lo operands (22 bits): the slot side of mode "fp16x2"
extern "C" int svps_slot_self_attn_f16(const float* qkv, float* out, int T, int L, int nheads, int head_dim, void* stream_) {
    return launch_self_attn<_Float16>(qkv, out, T, L, nheads, head_dim, stream_);
}

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

extern "C" int svps_row_softmax(const float* x, float* y, int rows, int cols, void* stream_) {
    if (!x || !y) return SVPS_ERR_BAD_ARG;
    if (rows <= 0 || cols <= 0) return SVPS_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(svps::row_softmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream_), x, y, rows, cols, 1.f);
    return (int)hipGetLastError();
}

extern "C" int svps_row_softmax_scaled(const float* x, float* y, int rows, int cols, float scale, void* stream_) {
    if (!x || !y) return SVPS_ERR_BAD_ARG;
    if (rows <= 0 || cols <= 0 || !(scale > 0.f)) return SVPS_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(svps::row_softmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream_), x, y, rows, cols, scale);
    return (int)hipGetLastError();
}
