// K9 - small batched products of the slot side with fp32-class precision, any strides, for gfx950.
//
// What is left of the slot update after K8 (dense layers with packed weights) are products whose B operand is an ACTIVATION,
// different per frame or clip, or whose shape K8 does not take:
//   the slot <-> slot retriever of the temporal head (mmdet/models/detectors/dynamic_mask_head.py:550-572):
//       logits^T = k q^T  [500 x 500 x 256 per clip],  out = softmax(.)^T v  [500 x 256 x 500 per clip]
//   the separable position terms of the fused retriever (csrc/retr_attn.hip):  cy = ytab Q''[:, :128]^T + a',  cx = xtab Q''[:, 128:]^T
//   the class projection (dynamic_mask_head.py:398, 256 -> 20 columns)
// The reference runs them in fp32. Here
//     C[b, m, n] = alpha * sum_k A[b, m, k] B[b, n, k]  (+ bias[b, n])
// with element strides for every index of A, B, C and bias (a transposed operand is a stride pattern, batch stride 0 = shared
// operand), on v_mfma_f32_32x32x16_bf16 with BOTH operands split on the fly into bf16 hi + lo and the three significant
// products accumulated in fp32 (the arithmetic of K8: relative error ~1e-5 of the largest term).
//
// Mapping: workgroup = 64 x 64 outputs of one batch entry, 4 waves = 2 x 2 blocks of 32 x 32; K in chunks of 32; per chunk the
// workgroup gathers A[64, 32] and B[64, 32] (16-byte loads along the contiguous index of each operand where alignment allows,
// strided scalar loads otherwise), splits them and stages hi / lo in LDS (80-byte rows: conflict-free 16-byte fragment reads), double-buffered, one
// barrier per chunk. These launches are microseconds of latency each (2 - 8 GFLOP per step in total): the kernel is kept
// simple and general, it is not a roofline kernel.
#include "common.h"
#include "../../include/slotvps_hip.h"

namespace svps {

constexpr int kBgTile = 64;                  // rows / columns per workgroup
constexpr int kBgK = 32;                     // k per chunk
constexpr int kBgRow = kBgK * 2 + 16;        // bytes per staged row (32 bf16 + pad)

struct BgArgs {
    const float* a;
    const float* b;
    const float* bias;
    float* c;
    long long sab, sam, sak;                 // A[b, m, k] at a + b * sab + m * sam + k * sak (elements)
    long long sbb, sbn, sbk;                 // B[b, n, k]
    long long scb, scm, scn;                 // C[b, m, n]
    long long sbias_b, sbias_n;              // bias[b, n]
    int M, N, K;
    float alpha;
};

// ST: the 16-bit type both operands are split into, hi + lo: __bf16 (16 bits of mantissa, fp32's range) or _Float16 (22 bits; |x| <
// 65 504, absolute resolution 6e-8 below 6.1e-5: svps_bgemm_f16, the position terms of the fused retriever)
template <typename ST>
__global__ __launch_bounds__(256) void bgemm_kernel(BgArgs g) {
    typedef ST st_x8 __attribute__((ext_vector_type(8)));
    typedef ST st_x4 __attribute__((ext_vector_type(4)));
    typedef ST st_x2 __attribute__((ext_vector_type(2)));
    // [buffer][operand A / B][hi / lo][64 rows][80 B]
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * 2 * kBgTile * kBgRow];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wm = w & 1, wn = w >> 1;
    const int m0 = blockIdx.x * kBgTile, n0 = blockIdx.y * kBgTile, bi = blockIdx.z;
    const float* A = g.a + (long long)bi * g.sab;
    const float* B = g.b + (long long)bi * g.sbb;
    const int nch = (g.K + kBgK - 1) / kBgK;

    // element e of a 64 x 32 operand tile -> (row, k): consecutive threads follow the operand's contiguous index
    const bool a_kc = g.sak == 1 || g.sam != 1, b_kc = g.sbk == 1 || g.sbn != 1;
    auto gather = [&](const float* P, long long srow, long long sk, int row0, int rows, bool kc, int ch, float (&v)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i;
            const int row = kc ? (e >> 5) : (e & 63), k = kc ? (e & 31) : (e >> 6);
            const int gr = row0 + row, gk = ch * kBgK + k;
            v[i] = (gr < rows && gk < g.K) ? P[(long long)gr * srow + (long long)gk * sk] : 0.f;
        }
    };
    auto split_store = [&](int buf, int op, bool kc, const float (&v)[8]) {
        char* base = smem + ((buf * 2 + op) * 2) * kBgTile * kBgRow;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i;
            const int row = kc ? (e >> 5) : (e & 63), k = kc ? (e & 31) : (e >> 6);
            const ST hi = (ST)v[i];
            const ST lo = (ST)(v[i] - (float)hi);
            *reinterpret_cast<ST*>(base + row * kBgRow + k * 2) = hi;
            *reinterpret_cast<ST*>(base + kBgTile * kBgRow + row * kBgRow + k * 2) = lo;
        }
    };

    // 16-byte forms of the two (64 rows x 32 k, 8 elements per thread):
    //   k contiguous (stride 1):   thread -> (row = e >> 3, k = 4 (e & 7)), e = tid, tid + 256: two float4 along k, two 8-byte LDS stores each (hi, lo)
    //   row contiguous (stride 1): thread -> (k = 2 (tid >> 4) + i, rows 4 (tid & 15) ..): two float4 along the rows, stored as (k, k + 1) pairs per row
    // used when every address involved is 16-byte aligned and the sizes are multiples of 4; anything else takes the scalar form
    auto vec_ok = [&](const float* P, long long sbatch, long long srow, long long sk, int rows) {
        const bool kcv = sk == 1, rcv = srow == 1;
        if (!kcv && !rcv) return 0;
        const long long sother = kcv ? srow : sk;
        if ((reinterpret_cast<uintptr_t>(P) & 15) || (sother & 3) || (sbatch & 3) || (rows & 3) || (g.K & 3)) return 0;
        return kcv ? 1 : 2;
    };
    const int a_vec = vec_ok(g.a, g.sab, g.sam, g.sak, g.M), b_vec = vec_ok(g.b, g.sbb, g.sbn, g.sbk, g.N);
    auto gather4 = [&](const float* P, long long srow, long long sk, int row0, int rows, int mode, int ch, float (&v)[8]) {
        if (mode == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = tid + 256 * i, row = e >> 3, k = 4 * (e & 7);
                const int gr = row0 + row, gk = ch * kBgK + k;
                f32x4 x = {0.f, 0.f, 0.f, 0.f};
                if (gr < rows && gk < g.K) x = *reinterpret_cast<const f32x4*>(P + (long long)gr * srow + gk);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[4 * i + j] = x[j];
            }
        } else {
            const int kq = tid >> 4, rq = tid & 15;
#pragma unroll
            for (int i = 0; i < 2; ++i) {                            // two float4 (rows 4 rq ..) for k = 2 kq + i
                const int k = 2 * kq + i, gr = row0 + 4 * rq, gk = ch * kBgK + k;
                f32x4 x = {0.f, 0.f, 0.f, 0.f};
                if (gr < rows && gk < g.K) x = *reinterpret_cast<const f32x4*>(P + (long long)gk * sk + gr);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[4 * i + j] = x[j];
            }
        }
    };
    auto split_store4 = [&](int buf, int op, int mode, const float (&v)[8]) {
        char* base = smem + ((buf * 2 + op) * 2) * kBgTile * kBgRow;
        if (mode == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = tid + 256 * i, row = e >> 3, k = 4 * (e & 7);
                st_x4 hi, lo;
#pragma unroll
                for (int j = 0; j < 4; ++j) { hi[j] = (ST)v[4 * i + j]; lo[j] = (ST)(v[4 * i + j] - (float)hi[j]); }
                *reinterpret_cast<st_x4*>(base + row * kBgRow + k * 2) = hi;
                *reinterpret_cast<st_x4*>(base + kBgTile * kBgRow + row * kBgRow + k * 2) = lo;
            }
        } else {
            const int kq = tid >> 4, rq = tid & 15;
#pragma unroll
            for (int j = 0; j < 4; ++j) {                            // row 4 rq + j: the pair (k = 2 kq, 2 kq + 1) as one 4-byte store
                st_x2 hi, lo;
#pragma unroll
                for (int i = 0; i < 2; ++i) { hi[i] = (ST)v[4 * i + j]; lo[i] = (ST)(v[4 * i + j] - (float)hi[i]); }
                *reinterpret_cast<st_x2*>(base + (4 * rq + j) * kBgRow + (2 * kq) * 2) = hi;
                *reinterpret_cast<st_x2*>(base + kBgTile * kBgRow + (4 * rq + j) * kBgRow + (2 * kq) * 2) = lo;
            }
        }
    };
    auto load_a = [&](int ch, float (&v)[8]) { if (a_vec) gather4(A, g.sam, g.sak, m0, g.M, a_vec, ch, v); else gather(A, g.sam, g.sak, m0, g.M, a_kc, ch, v); };
    auto load_b = [&](int ch, float (&v)[8]) { if (b_vec) gather4(B, g.sbn, g.sbk, n0, g.N, b_vec, ch, v); else gather(B, g.sbn, g.sbk, n0, g.N, b_kc, ch, v); };
    auto put_a = [&](int buf, const float (&v)[8]) { if (a_vec) split_store4(buf, 0, a_vec, v); else split_store(buf, 0, a_kc, v); };
    auto put_b = [&](int buf, const float (&v)[8]) { if (b_vec) split_store4(buf, 1, b_vec, v); else split_store(buf, 1, b_kc, v); };

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float av[8], bv[8];
    load_a(0, av);
    load_b(0, bv);
    put_a(0, av);
    put_b(0, bv);
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nch) {                                        // the next chunk's loads fly under this chunk's MFMAs
            load_a(ch + 1, av);
            load_b(ch + 1, bv);
        }
        const char* ah = smem + ((buf * 2 + 0) * 2) * kBgTile * kBgRow + (32 * wm + r) * kBgRow + 16 * h;
        const char* bh = smem + ((buf * 2 + 1) * 2) * kBgTile * kBgRow + (32 * wn + r) * kBgRow + 16 * h;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const st_x8 xh = *reinterpret_cast<const st_x8*>(ah + 32 * u);
            const st_x8 xl = *reinterpret_cast<const st_x8*>(ah + kBgTile * kBgRow + 32 * u);
            const st_x8 yh = *reinterpret_cast<const st_x8*>(bh + 32 * u);
            const st_x8 yl = *reinterpret_cast<const st_x8*>(bh + kBgTile * kBgRow + 32 * u);
            if constexpr (sizeof(ST) == 2 && __is_same(ST, _Float16)) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, yh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yl, acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, yh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yl, acc, 0, 0, 0);
            }
        }
        if (ch + 1 < nch) {
            put_a(buf ^ 1, av);
            put_b(buf ^ 1, bv);
        }
        __syncthreads();
    }

    // epilogue: register i = row (i & 3) + 8 (i >> 2) + 4 h of the wave's block, column = lane r
    const int n = n0 + 32 * wn + r;
    if (n < g.N) {
        const float bvn = g.bias ? g.bias[(long long)bi * g.sbias_b + (long long)n * g.sbias_n] : 0.f;
        float* C = g.c + (long long)bi * g.scb + (long long)n * g.scn;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = m0 + 32 * wm + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (m < g.M) C[(long long)m * g.scm] = g.alpha * acc[i] + bvn;
        }
    }
}

}  // namespace svps

namespace {
template <typename ST>
int bgemm_launch(const float* a, const long long* sa, const float* b, const long long* sb, const float* bias,
                 const long long* sbias, float* c, const long long* sc, int batch, int M, int N, int K, float alpha,
                 void* stream_) {
    if (!a || !b || !c || !sa || !sb || !sc || (bias && !sbias)) return SVPS_ERR_BAD_ARG;
    if (batch <= 0 || M <= 0 || N <= 0 || K <= 0 || batch > 65535) return SVPS_ERR_BAD_SHAPE;
    svps::BgArgs g;
    g.a = a; g.b = b; g.bias = bias; g.c = c;
    g.sab = sa[0]; g.sam = sa[1]; g.sak = sa[2];
    g.sbb = sb[0]; g.sbn = sb[1]; g.sbk = sb[2];
    g.scb = sc[0]; g.scm = sc[1]; g.scn = sc[2];
    g.sbias_b = bias ? sbias[0] : 0; g.sbias_n = bias ? sbias[1] : 0;
    g.M = M; g.N = N; g.K = K; g.alpha = alpha;
    const dim3 grid((M + svps::kBgTile - 1) / svps::kBgTile, (N + svps::kBgTile - 1) / svps::kBgTile, batch);
    if (grid.y > 65535) return SVPS_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(svps::bgemm_kernel<ST>, grid, dim3(256), 0, static_cast<hipStream_t>(stream_), g);
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int svps_bgemm(const float* a, const long long* sa, const float* b, const long long* sb, const float* bias,
                          const long long* sbias, float* c, const long long* sc, int batch, int M, int N, int K, float alpha,
                          void* stream_) {
    return bgemm_launch<__bf16>(a, sa, b, sb, bias, sbias, c, sc, batch, M, N, K, alpha, stream_);
}

extern "C" int svps_bgemm_f16(const float* a, const long long* sa, const float* b, const long long* sb, const float* bias,
                              const long long* sbias, float* c, const long long* sc, int batch, int M, int N, int K, float alpha,
                              void* stream_) {
    return bgemm_launch<_Float16>(a, sa, b, sb, bias, sbias, c, sc, batch, M, N, K, alpha, stream_);
}
