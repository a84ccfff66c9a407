// K8 - the dense layers of the slot update on the matrix cores with fp32-class precision, for gfx950.
//
// The slot side of a stage (mmdet/models/detectors/dynamic_mask_head.py:342-400, :494-572) is a chain of nn.Linear layers on
// [T * L, 256] rows: self-attention projections, to_q, the FFN (256 -> 2048 -> 256), the temporal head, the class / embedding
// towers. The reference runs them in fp32; the fp32 matrix instructions of gfx950 run at the vector rate (1/16 of bf16), and
// with 8 000 rows per launch the GEMM library reaches ~50 TFLOP/s on them. Here
//     y[m, n] = act( sum_k x[m, k] W[n, k] + b[n] )
// runs on v_mfma_f32_32x32x16_bf16 with BOTH operands carried as bf16 hi + lo (16-bit mantissa) and the three significant
// products accumulated in fp32 ("split-bf16": relative error ~1e-5 of the largest term, the class of fp32 summation-order
// noise), the weight pre-packed by the host into MFMA B-fragment order (hi and lo), the activation rows split on the fly.
//
// Mapping: workgroup = 64 rows x 256 output columns, 8 waves = 2 row blocks x 4 column groups of 64 (two 32 x 32
// accumulators per wave). K is walked in chunks of 64: the workgroup stages x[64 rows, 64 k] as bf16 hi / lo in LDS
// (double-buffered, one barrier per chunk), every wave reads its A fragments from there and streams its B fragments from
// L2 in fragment order (1 KiB per wave instruction, each byte of the weight once per workgroup).
// Epilogue: + bias[n], ReLU or GELU (exact erf form, F.gelu's default), fp32 store (32 consecutive columns per half-wave).
// LN epilogue (svps_slot_gemm_ln, N = 256: the workgroup holds whole rows): the tile goes through LDS instead of HBM and the
// step that follows most dense layers of the slot update,  y = LN(gemm [+ pre]) * gamma + beta (+ReLU) (+post), runs in the
// same launch - one wave per row with the arithmetic of svps_row_ln, operation for operation (results bitwise equal to the
// two-launch form).
#include "common.h"
#include "../../include/slotvps_hip.h"

namespace svps {

constexpr int kGmRows = 64;                 // rows per workgroup (template RBW = 2; RBW = 1: 32 rows, for launches with few workgroups)
constexpr int kGmCols = 256;                // output columns per workgroup
constexpr int kGmK = 64;                    // k per chunk
constexpr int kGmRow = kGmK * 2 + 16;       // bytes per row of the staged A tile (padded: conflict-free 16-byte fragment reads)

struct GemmLds {
    static constexpr int buf_bytes = 2 * kGmRows * kGmRow;      // hi | lo
    static constexpr int nbuf = 4;                              // K <= 256: every chunk of the A tile is staged up front
    static constexpr int total = nbuf * buf_bytes;
    static constexpr int ln_row = kGmCols * 4 + 16;            // LN epilogue: fp32 rows of the output tile, padded (bank spread of the two lane halves)
};
static_assert(kGmRows * GemmLds::ln_row <= GemmLds::total, "the LN epilogue reuses the staging buffers");

struct GemmLn {                              // arguments of the LN epilogue
    const float* pre;                        // [M, 256] added before the LayerNorm, or null
    const float* post;                       // [M, 256] added after it, or null
    const float* gamma;
    const float* beta;
    float eps;
    int relu;
};

__device__ __forceinline__ float gm_wave_sum(float x) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m);
    return x;
}


// ACT: 0 none, 1 ReLU, 2 GELU (erf); RBW: row blocks per workgroup (2 or 1); PREF: weights of the next chunk requested before
// this chunk's MFMAs (register double buffer, 192 registers: one workgroup per CU - for launches that do not fill the chip
// several times over; wide layers run several 110-register workgroups per CU instead, which hide the latency by themselves)
// F16: both operands split into fp16 hi + lo instead of bf16 hi + lo (22 instead of 16 bits of mantissa for the same three MFMAs;
// fp16's range: |x| < 65 504, absolute resolution 6e-8 below 6.1e-5) - the query side of the fused retriever (to_q, the key fold),
// whose operands are LayerNorm outputs and O(0.1) weights and whose result goes through logits that cancel (svps_slot_gemm_f16)
typedef __attribute__((ext_vector_type(8))) _Float16 gm_f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 gm_f16x4;

template <int ACT, int RBW, bool PREF, bool LN = false, bool F16 = false>
__global__ __launch_bounds__(512) void slot_gemm_kernel(const float* __restrict__ x,        // [M, K]
                                                        const __bf16* __restrict__ wpack,   // [N/32][K/16][2][64][8]
                                                        const float* __restrict__ bias,     // [N] or null
                                                        float* __restrict__ y,              // [M, N]
                                                        int M, int K, int N, GemmLn ln = GemmLn{}) {
    __shared__ __attribute__((aligned(16))) char smem[GemmLds::total];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    constexpr int ROWS = 32 * RBW, NB = RBW;                    // rows per workgroup; column blocks per wave (8 waves cover 8 x RBW blocks)
    const int rb = w % RBW, cg = w / RBW;                       // row block, column group (32 NB columns)
    const int m0 = blockIdx.x * ROWS, n0 = blockIdx.y * kGmCols;
    const int KS = K / 16, nch = (K + kGmK - 1) / kGmK;

    // A tile staging: thread -> RBW (row, 4-k group) items per chunk
    auto gather = [&](int ch, f32x4 (&av)[RBW]) {
#pragma unroll
        for (int i = 0; i < RBW; ++i) {
            const int q = tid + 512 * i, row = q >> 4, kg = (q & 15) * 4;
            const int m = m0 + row, k = ch * kGmK + kg;
            av[i] = (m < M && k < K) ? *reinterpret_cast<const f32x4*>(x + (size_t)m * K + k) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto split_store = [&](int buf, const f32x4 (&av)[RBW]) {
        char* bh = smem + buf * GemmLds::buf_bytes;
#pragma unroll
        for (int i = 0; i < RBW; ++i) {
            const int q = tid + 512 * i, row = q >> 4, kg = (q & 15) * 4;
            if constexpr (F16) {
                gm_f16x4 vh, vl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    vh[e] = (_Float16)av[i][e];
                    vl[e] = (_Float16)(av[i][e] - (float)vh[e]);
                }
                *reinterpret_cast<gm_f16x4*>(bh + row * kGmRow + kg * 2) = vh;
                *reinterpret_cast<gm_f16x4*>(bh + kGmRows * kGmRow + row * kGmRow + kg * 2) = vl;
            } else {
                bf16x4 vh, vl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    vh[e] = (__bf16)av[i][e];
                    vl[e] = (__bf16)(av[i][e] - (float)vh[e]);
                }
                *reinterpret_cast<bf16x4*>(bh + row * kGmRow + kg * 2) = vh;
                *reinterpret_cast<bf16x4*>(bh + kGmRows * kGmRow + row * kGmRow + kg * 2) = vl;
            }
        }
    };

    f32x16 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;

    // B fragments of column block cb, k-step ks, part p: u32x4 index ((cb * KS + ks) * 2 + p) * 64 + lane
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(wpack) + lane;
    const int cb0 = (n0 >> 5) + NB * cg;
    // weights of one chunk: 4 k-steps x NB column blocks x (hi, lo) = 8 NB fragments; with PREF the NEXT chunk's set is
    // requested before this chunk's MFMAs (register double buffer), so that the L2 latency of the weight stream (one
    // workgroup per CU for the 256-column layers: nobody else to hide it) runs under the matrix work
    auto load_w = [&](int ch, u32x4 (&wb)[8 * NB]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ks = ch * 4 + u;
            const int kc = ks < KS ? ks : KS - 1;               // short last chunk: a valid address, the fragment is not used
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const size_t f = ((size_t)(cb0 + b) * KS + kc) * 2 * 64;
                wb[2 * NB * u + 2 * b] = wsrc[f];
                wb[2 * NB * u + 2 * b + 1] = wsrc[f + 64];
            }
        }
    };
    auto mma = [&](int ch, int buf, const u32x4 (&wb)[8 * NB]) {
        const char* ah = smem + buf * GemmLds::buf_bytes + (32 * rb + r) * kGmRow + 16 * h;
        const char* al = ah + kGmRows * kGmRow;
        const int nks = (KS - ch * 4) < 4 ? (KS - ch * 4) : 4;   // k-steps of this chunk (the last chunk may be short)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (u < nks) {
                if constexpr (F16) {
                    const gm_f16x8 xh = *reinterpret_cast<const gm_f16x8*>(ah + 32 * u);
                    const gm_f16x8 xl = *reinterpret_cast<const gm_f16x8*>(al + 32 * u);
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const gm_f16x8 wh = __builtin_bit_cast(gm_f16x8, wb[2 * NB * u + 2 * b]);
                        const gm_f16x8 wl = __builtin_bit_cast(gm_f16x8, wb[2 * NB * u + 2 * b + 1]);
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wh, acc[b], 0, 0, 0);
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, wh, acc[b], 0, 0, 0);
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wl, acc[b], 0, 0, 0);
                    }
                } else {
                    const bf16x8 xh = *reinterpret_cast<const bf16x8*>(ah + 32 * u);
                    const bf16x8 xl = *reinterpret_cast<const bf16x8*>(al + 32 * u);
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const bf16x8 wh = __builtin_bit_cast(bf16x8, wb[2 * NB * u + 2 * b]);
                        const bf16x8 wl = __builtin_bit_cast(bf16x8, wb[2 * NB * u + 2 * b + 1]);
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, wh, acc[b], 0, 0, 0);
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, wh, acc[b], 0, 0, 0);
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, wl, acc[b], 0, 0, 0);
                    }
                }
            }
        }
    };
    u32x4 w0[8 * NB], w1[PREF ? 8 * NB : 1];
    if constexpr (!PREF) {
        // ---- wide layers (more workgroups than CUs, two co-resident per CU hide each other's latencies; 128 registers at most):
        // two LDS buffers, the A rows of the next chunk requested after this chunk's MFMAs
        f32x4 av[RBW];
        gather(0, av);
        split_store(0, av);
        __syncthreads();
        for (int ch = 0; ch < nch; ++ch) {
            load_w(ch, w0);
            mma(ch, ch & 1, w0);
            if (ch + 1 < nch) {
                gather(ch + 1, av);
                split_store((ch + 1) & 1, av);
            }
            __syncthreads();
        }
    } else if (nch <= GemmLds::nbuf) {
        // ---- K <= 256 (every 256-wide layer of the slot side): the whole A tile is requested at once and staged before the
        // first MFMA - ONE global-memory latency and ONE barrier per launch instead of one of each per chunk (these launches
        // are a few microseconds of pure latency: 32 x 256 x 256 per workgroup is 48 MFMAs per wave)
        // (64-row tiles: in two groups of two chunks, 16 registers instead of 32 - the wide layers need two workgroups per CU)
        constexpr int GS = RBW == 1 ? GemmLds::nbuf : GemmLds::nbuf / 2;
        load_w(0, w0);
#pragma unroll
        for (int g0 = 0; g0 < GemmLds::nbuf; g0 += GS) {
            f32x4 av[GS][RBW];
#pragma unroll
            for (int ch = 0; ch < GS; ++ch)
                if (g0 + ch < nch) gather(g0 + ch, av[ch]);
#pragma unroll
            for (int ch = 0; ch < GS; ++ch)
                if (g0 + ch < nch) split_store(g0 + ch, av[ch]);
        }
        __syncthreads();
#pragma unroll
        for (int ch = 0; ch < GemmLds::nbuf; ch += 2) {
            if (ch >= nch) break;
            if constexpr (PREF) {
                if (ch + 1 < nch) load_w(ch + 1, w1);
                mma(ch, ch, w0);
                if (ch + 1 < nch) {
                    if (ch + 2 < nch) load_w(ch + 2, w0);
                    mma(ch + 1, ch + 1, w1);
                }
            } else {
                mma(ch, ch, w0);
                if (ch + 1 < nch) { load_w(ch + 1, w0); mma(ch + 1, ch + 1, w0); }
                if (ch + 2 < nch) load_w(ch + 2, w0);
            }
        }
    } else {
        // ---- long K (the second FFN layer, K = 2048): two LDS buffers, the A rows of chunk ch+2 are requested while chunk ch is on
        // the matrix cores and split into LDS one iteration later (a request issued one chunk ahead was still in flight when
        // it was needed: every chunk paid a global-memory latency)
        f32x4 ga[2][RBW];
        gather(0, ga[0]);
        load_w(0, w0);
        gather(1, ga[1]);
        split_store(0, ga[0]);
        if (2 < nch) gather(2, ga[0]);
        __syncthreads();
        for (int ch = 0; ch < nch; ch += 2) {
            // even chunk ch: A rows in buffer 0, weights in w0; registers: ga[1] = chunk ch+1, ga[0] = chunk ch+2
            if constexpr (PREF) { if (ch + 1 < nch) load_w(ch + 1, w1); }
            mma(ch, 0, w0);
            if (ch + 1 < nch) split_store(1, ga[1]);
            if (ch + 3 < nch) gather(ch + 3, ga[1]);
            __syncthreads();
            if (ch + 1 >= nch) break;
            // odd chunk ch+1: buffer 1, weights in w1 (PREF) or reloaded into w0
            if constexpr (PREF) {
                if (ch + 2 < nch) load_w(ch + 2, w0);
                mma(ch + 1, 1, w1);
            } else {
                load_w(ch + 1, w0);
                mma(ch + 1, 1, w0);
                if (ch + 2 < nch) load_w(ch + 2, w0);
            }
            if (ch + 2 < nch) split_store(0, ga[0]);
            if (ch + 4 < nch) gather(ch + 4, ga[0]);
            __syncthreads();
        }
    }

    if constexpr (LN) {
        // ---- LN epilogue (N == 256, ACT == 0): tile -> LDS, then one wave per row exactly as svps_row_ln does it
        // the residual rows and the affine pair are requested FIRST: their latency (HBM for `pre`) runs under the LDS transpose
        constexpr int RPW = ROWS / 8;                           // rows per wave
        float4 pv[RPW], qv[RPW];
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const int m = m0 + w + 8 * j;
            const size_t base = (size_t)(m < M ? m : M - 1) * 256 + 4 * lane;
            pv[j] = ln.pre ? *reinterpret_cast<const float4*>(ln.pre + base) : make_float4(0.f, 0.f, 0.f, 0.f);
            qv[j] = ln.post ? *reinterpret_cast<const float4*>(ln.post + base) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float4 ww = *reinterpret_cast<const float4*>(ln.gamma + 4 * lane);
        const float4 bb = *reinterpret_cast<const float4*>(ln.beta + 4 * lane);
        __syncthreads();                                        // every wave is done with the staged A tile
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int col = 32 * NB * cg + 32 * b + r;
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = 32 * rb + (i & 3) + 8 * (i >> 2) + 4 * h;
                *reinterpret_cast<float*>(smem + row * GemmLds::ln_row + col * 4) = acc[b][i] + bv;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const int row = w + 8 * j, m = m0 + row;
            if (m >= M) break;
            float4 v = *reinterpret_cast<const float4*>(smem + row * GemmLds::ln_row + 16 * lane);
            if (ln.pre) { v.x += pv[j].x; v.y += pv[j].y; v.z += pv[j].z; v.w += pv[j].w; }
            const float mean = gm_wave_sum(v.x + v.y + v.z + v.w) * (1.f / 256.f);
            const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
            const float var = gm_wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.f / 256.f);
            const float rstd = rsqrtf(var + ln.eps);
            float4 o = make_float4(dx * rstd * ww.x + bb.x, dy * rstd * ww.y + bb.y, dz * rstd * ww.z + bb.z, dw * rstd * ww.w + bb.w);
            if (ln.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
            if (ln.post) { o.x += qv[j].x; o.y += qv[j].y; o.z += qv[j].z; o.w += qv[j].w; }
            *reinterpret_cast<float4*>(y + (size_t)m * 256 + 4 * lane) = o;
        }
        return;
    }
    // ---- epilogue: register i of a block = row (i & 3) + 8 (i >> 2) + 4 h, column = lane r
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int n = n0 + 32 * NB * cg + 32 * b + r;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = m0 + 32 * rb + (i & 3) + 8 * (i >> 2) + 4 * h;
            float v = acc[b][i] + bv;
            if constexpr (ACT == 1) v = v > 0.f ? v : 0.f;
            if constexpr (ACT == 2) v = svps_gelu_erf(v);
            if (m < M) y[(size_t)m * N + n] = v;
        }
    }
}

}  // namespace svps

extern "C" int svps_slot_gemm(const float* x, const void* wpack, const float* bias, float* y, int M, int K, int N, int act,
                              void* stream_) {
    if (!x || !wpack || !y) return SVPS_ERR_BAD_ARG;
    if (M <= 0 || K <= 0 || (K & 15) || N <= 0 || (N % svps::kGmCols) || act < 0 || act > 2) return SVPS_ERR_BAD_SHAPE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const __bf16* wp = static_cast<const __bf16*>(wpack);
    // 32-row tiles (112 registers: two workgroups per CU hide each other's latencies) unless 64-row tiles already give every CU two
    // workgroups: the 256-column layers on 16 000 rows are 250 tiles of 64 rows - one per CU, a launch of pure latency (27 us
    // against 20 with 500 tiles of 32 rows; step 37.7 -> 37.2 ms)
    const int wg64 = ((M + 63) / 64) * (N / svps::kGmCols);
    const bool small = wg64 < 2 * svps_num_cus();
    const int rows = small ? 32 : 64;
    const dim3 grid((M + rows - 1) / rows, N / svps::kGmCols);
#define SVPS_GEMM(A, R, P) hipLaunchKernelGGL((svps::slot_gemm_kernel<A, R, P>), grid, dim3(512), 0, stream, x, wp, bias, y, M, K, N)
    if (small) {                                               // (not small = at least two 64-row workgroups per CU: the light variant)
        if (act == 0) SVPS_GEMM(0, 1, true); else if (act == 1) SVPS_GEMM(1, 1, true); else SVPS_GEMM(2, 1, true);
    } else {
        if (act == 0) SVPS_GEMM(0, 2, false); else if (act == 1) SVPS_GEMM(1, 2, false); else SVPS_GEMM(2, 2, false);
    }
#undef SVPS_GEMM
    return (int)hipGetLastError();
}

extern "C" int svps_slot_gemm_f16(const float* x, const void* wpack, const float* bias, float* y, int M, int K, int N, void* stream_) {
    if (!x || !wpack || !y) return SVPS_ERR_BAD_ARG;
    if (M <= 0 || K <= 0 || (K & 15) || N <= 0 || (N % svps::kGmCols)) return SVPS_ERR_BAD_SHAPE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const __bf16* wp = static_cast<const __bf16*>(wpack);      // raw 16-bit words: fp16 hi / lo fragments (ops.pack_b_fragments(..., split="fp16"))
    const int wg64 = ((M + 63) / 64) * (N / svps::kGmCols);
    const bool small = wg64 < 2 * svps_num_cus();
    const int rows = small ? 32 : 64;
    const dim3 grid((M + rows - 1) / rows, N / svps::kGmCols);
    if (small) hipLaunchKernelGGL((svps::slot_gemm_kernel<0, 1, true, false, true>), grid, dim3(512), 0, stream, x, wp, bias, y, M, K, N);
    else hipLaunchKernelGGL((svps::slot_gemm_kernel<0, 2, false, false, true>), grid, dim3(512), 0, stream, x, wp, bias, y, M, K, N);
    return (int)hipGetLastError();
}

// the fp16-split form with the activation / LayerNorm epilogues of the bf16-split form (round 4: the slot side of precision "fp16x2"
// ran its activations and LayerNorms as separate launches)
extern "C" int svps_slot_gemm_f16_act(const float* x, const void* wpack, const float* bias, float* y, int M, int K, int N, int act,
                                      void* stream_) {
    if (!x || !wpack || !y) return SVPS_ERR_BAD_ARG;
    if (M <= 0 || K <= 0 || (K & 15) || N <= 0 || (N % svps::kGmCols) || act < 0 || act > 2) return SVPS_ERR_BAD_SHAPE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const __bf16* wp = static_cast<const __bf16*>(wpack);
    const int wg64 = ((M + 63) / 64) * (N / svps::kGmCols);
    const bool small = wg64 < 2 * svps_num_cus();
    const int rows = small ? 32 : 64;
    const dim3 grid((M + rows - 1) / rows, N / svps::kGmCols);
#define SVPS_GEMM(A, R, P) hipLaunchKernelGGL((svps::slot_gemm_kernel<A, R, P, false, true>), grid, dim3(512), 0, stream, x, wp, bias, y, M, K, N)
    if (small) {
        if (act == 0) SVPS_GEMM(0, 1, true); else if (act == 1) SVPS_GEMM(1, 1, true); else SVPS_GEMM(2, 1, true);
    } else {
        if (act == 0) SVPS_GEMM(0, 2, false); else if (act == 1) SVPS_GEMM(1, 2, false); else SVPS_GEMM(2, 2, false);
    }
#undef SVPS_GEMM
    return (int)hipGetLastError();
}

extern "C" int svps_slot_gemm_ln_f16(const float* x, const void* wpack, const float* bias, const float* pre, const float* post,
                                     const float* gamma, const float* beta, float eps, int relu, float* y, int M, int K,
                                     void* stream_) {
    if (!x || !wpack || !y || !gamma || !beta) return SVPS_ERR_BAD_ARG;
    if (M <= 0 || K <= 0 || (K & 15)) return SVPS_ERR_BAD_SHAPE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const __bf16* wp = static_cast<const __bf16*>(wpack);
    const int N = svps::kGmCols;
    const svps::GemmLn ln{pre, post, gamma, beta, eps, relu};
    const int wg64 = (M + 63) / 64;
    const bool small = wg64 < 2 * svps_num_cus();
    const int rows = small ? 32 : 64;
    const dim3 grid((M + rows - 1) / rows, 1);
    if (small) hipLaunchKernelGGL((svps::slot_gemm_kernel<0, 1, true, true, true>), grid, dim3(512), 0, stream, x, wp, bias, y, M, K, N, ln);
    else hipLaunchKernelGGL((svps::slot_gemm_kernel<0, 2, false, true, true>), grid, dim3(512), 0, stream, x, wp, bias, y, M, K, N, ln);
    return (int)hipGetLastError();
}

extern "C" int svps_slot_gemm_ln(const float* x, const void* wpack, const float* bias, const float* pre, const float* post,
                                 const float* gamma, const float* beta, float eps, int relu, float* y, int M, int K,
                                 void* stream_) {
    if (!x || !wpack || !y || !gamma || !beta) return SVPS_ERR_BAD_ARG;
    if (M <= 0 || K <= 0 || (K & 15)) return SVPS_ERR_BAD_SHAPE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const __bf16* wp = static_cast<const __bf16*>(wpack);
    const int N = svps::kGmCols;
    const svps::GemmLn ln{pre, post, gamma, beta, eps, relu};
    const int wg64 = (M + 63) / 64;
    const bool small = wg64 < 2 * svps_num_cus();
    const int rows = small ? 32 : 64;
    const dim3 grid((M + rows - 1) / rows, 1);
#define SVPS_GEMM_LN(R, P) hipLaunchKernelGGL((svps::slot_gemm_kernel<0, R, P, true>), grid, dim3(512), 0, stream, x, wp, bias, y, M, K, N, ln)
    if (small) SVPS_GEMM_LN(1, true); else SVPS_GEMM_LN(2, false);
#undef SVPS_GEMM_LN
    return (int)hipGetLastError();
}
