// K3t - the LayerNorm statistics of the fused retriever with BOTH factors carried as FP16 hi + lo (the "tight" statistics mode).
//
// Error budget of the statistics-fused retriever against a float64 evaluation of MaskDynamicConv.forward
// (mmdet/models/detectors/dynamic_mask_head.py:423-461), measured by perturbing one source at a time: P * rstd_v
// carried as ONE fp16 8.7e-4 (retr_attn.hip has a hi + lo form for it), rstd_v from the FP16 value factor (~5e-5 relative) 2e-4 -
// 4e-4, rstd_k from the FP16 key factor (~3e-5 relative, in front of logits of magnitude up to 80) 1e-4. K3' / K3'' (retr_stats.hip,
// retr_stats2.hip) round the QR factors to FP16, 11 bits per entry. Here a factor is R = hi + lo (two FP16 matrices, 22 bits): the
// products v_mfma_f32_32x32x16_f16 R_hi x and R_lo x run into ONE fp32 accumulator, everything else as in K3' (f exact in FP16,
// position terms as fp32 tables in the accumulator's initial value, fp32 sums of squares, v_rsq_f32): rstd to ~2e-7.
//
// This is a PRECISION mode, not the fast path: a plain kernel (one barrier per tile, register-staged tiles, hipcc's schedule), launched
// once per projection (key, value: the map is read twice), one row block of the hi and lo factor resident per wave (128 registers).
// Output: the 16-byte aux rows of K3' (same layout, consumed by retr_attn.hip unchanged); the key launch writes bytes 8 .. 11, the
// value launch the rest.
#include "common.h"
#include "../../include/slotvps_hip.h"

namespace svps {

typedef __attribute__((ext_vector_type(8))) _Float16 st_f16x8;
typedef __attribute__((ext_vector_type(2))) uint32_t st_u32x2;

constexpr int kTtRow = 256 * 2 + 16;                 // staged pixel row: 256 fp16 + pad (conflict-free 16-byte fragment reads)
struct StatsTLds {
    static constexpr int xt = 0;                     // [2][32 px][528 B]
    static constexpr int x_bytes = kTilePx * kTtRow;
    static constexpr int part = 2 * x_bytes;         // [2][8 waves][32 px] float: sum of squares of the wave's 32 rows
    static constexpr int total = part + 2 * 8 * 32 * 4;
};

struct StatsTArgs {
    const __bf16* feat;        // [T, HW, 256]
    const float* ty;           // [H, 256] or null:  R[:, :128] pos_y[y]   (key launch only)
    const float* tx;           // [W, 256] or null:  R[:, 128:] pos_x[x]
    const _Float16* r_hi;      // [256, 256] upper triangular
    const _Float16* r_lo;      // [256, 256]  R - hi
    const float* rb;           // [256]  the column r of [R | r]
    __bf16* aux;               // [T, HW, 8]
    float eps;
    int HW, H, W, tiles_per_wg;
    int value;                 // 0: key launch (writes rstd_k), 1: value launch (writes {1, sigma_v hi, lo, 0} and rstd_v)
    int map_f16;               // the map is fp16 (staged as it is)
};

// one wave = row block RB (rows 32 RB .. 32 RB + 31) of the three factors; R is upper triangular: k-steps 2 RB .. 15 only
template <int RB>
__device__ __forceinline__ void stats_tight_role(const StatsTArgs& a, char* smem, int w, int lane) {
    constexpr int KS0 = 2 * RB, NKS = 16 - KS0;
    const int r = lane & 31, h = lane >> 5;
    const int t = blockIdx.y;
    st_f16x8 fh[NKS], fl[NKS];
    {
        const size_t row = (size_t)(32 * RB + r) * 256 + 8 * h;
#pragma unroll
        for (int i = 0; i < NKS; ++i) {
            fh[i] = *reinterpret_cast<const st_f16x8*>(a.r_hi + row + 16 * (KS0 + i));
            fl[i] = *reinterpret_cast<const st_f16x8*>(a.r_lo + row + 16 * (KS0 + i));
        }
    }
    // accumulator register 4 g + j <-> factor row 32 RB + 8 g + 4 h + j
    const int tid = threadIdx.x;
    const int tiles = (a.HW + kTilePx - 1) / kTilePx;
    const int tile0 = blockIdx.x * a.tiles_per_wg;
    int tile1 = tile0 + a.tiles_per_wg;
    tile1 = tile1 < tiles ? tile1 : tiles;
    const __bf16* F = a.feat + (size_t)t * a.HW * 256;
    float* part = reinterpret_cast<float*>(smem + StatsTLds::part);
    // staging: thread -> (pixel tid >> 4, 32 bytes = 16 channels); bf16 -> fp16 is exact for |f| in [6.1e-5, 65504]
    const int spx = tid >> 4, sc16 = tid & 15;
    bf16x8 v0, v1;
    auto fetch = [&](int tile) {
        int gp = tile * kTilePx + spx;
        gp = gp < a.HW ? gp : a.HW - 1;                              // ragged last tile / past the chunk: a valid pixel (not stored)
        v0 = *reinterpret_cast<const bf16x8*>(F + (size_t)gp * 256 + 16 * sc16);
        v1 = *reinterpret_cast<const bf16x8*>(F + (size_t)gp * 256 + 16 * sc16 + 8);
    };
    auto stage = [&](int buf) {
        st_f16x8 o0, o1;
#pragma unroll
        for (int j = 0; j < 8; ++j) { o0[j] = (_Float16)(float)v0[j]; o1[j] = (_Float16)(float)v1[j]; }
        if (a.map_f16) { o0 = __builtin_bit_cast(st_f16x8, v0); o1 = __builtin_bit_cast(st_f16x8, v1); }
        char* dst = smem + StatsTLds::xt + buf * StatsTLds::x_bytes + spx * kTtRow + 32 * sc16;
        *reinterpret_cast<st_f16x8*>(dst) = o0;
        *reinterpret_cast<st_f16x8*>(dst + 16) = o1;
    };
    fetch(tile0);
    stage(0);
    __syncthreads();
    // one barrier per tile: the next tile is fetched under this tile's MFMAs and staged into the other buffer behind them; wave 0
    // finishes a tile from the (double-buffered) partial sums after the barrier while the others go on
    for (int tile = tile0; tile < tile1; ++tile) {
        const int cur = (tile - tile0) & 1;
        const int px0 = tile * kTilePx;
        if (tile + 1 < tile1) fetch(tile + 1);
        // initial value: the column r (+ the projected position terms of this lane's pixel, key launch)
        f32x16 ak;
        {
            int gp = px0 + r;
            gp = gp < a.HW ? gp : a.HW - 1;
            const int y = gp / a.W, x = gp - y * a.W;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 p = *reinterpret_cast<const f32x4*>(a.rb + 32 * RB + 8 * g + 4 * h);
                if (a.ty) {
                    const f32x4 py = *reinterpret_cast<const f32x4*>(a.ty + (size_t)y * 256 + 32 * RB + 8 * g + 4 * h);
                    const f32x4 pxv = *reinterpret_cast<const f32x4*>(a.tx + (size_t)x * 256 + 32 * RB + 8 * g + 4 * h);
#pragma unroll
                    for (int j = 0; j < 4; ++j) p[j] += py[j] + pxv[j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) ak[4 * g + j] = p[j];
            }
        }
        const char* xrow = smem + StatsTLds::xt + cur * StatsTLds::x_bytes + r * kTtRow + 16 * h;
#pragma unroll
        for (int i = 0; i < NKS; ++i) {
            const st_f16x8 xf = *reinterpret_cast<const st_f16x8*>(xrow + 32 * (KS0 + i));
            ak = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[i], xf, ak, 0, 0, 0);
            ak = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[i], xf, ak, 0, 0, 0);
        }
        float sk = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) sk += ak[i] * ak[i];
        sk += __shfl_xor(sk, 32);
        if (h == 0) part[cur * 256 + w * 32 + r] = sk;
        if (tile + 1 < tile1) stage(cur ^ 1);
        __syncthreads();                                             // partial sums of this tile complete; the next tile is staged
        if (w == 0 && h == 0) {
            float tk = 0.f;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) tk += part[cur * 256 + ww * 32 + r];
            const float var = tk * (1.f / 256.f) + a.eps;
            const float rstd = __builtin_amdgcn_rsqf(var);
            const int gp = px0 + r;
            if (gp < a.HW) {
                char* row = reinterpret_cast<char*>(a.aux) + ((size_t)t * a.HW + gp) * 16;
                if (!a.value) {
                    *reinterpret_cast<float*>(row + 8) = rstd;
                } else {
                    float sigma = var * rstd;
                    asm volatile("" : "+v"(sigma));                  // one fp32 value for both halves (see retr_attn.hip, p2_store)
                    const _Float16 sh = (_Float16)sigma, sl = (_Float16)(sigma - (float)sh), one = (_Float16)1.0f;
                    const uint32_t w0 = (uint32_t)__builtin_bit_cast(uint16_t, one) | ((uint32_t)__builtin_bit_cast(uint16_t, sh) << 16);
                    const uint32_t w1 = (uint32_t)__builtin_bit_cast(uint16_t, sl);
                    *reinterpret_cast<st_u32x2*>(row) = st_u32x2{w0, w1};
                    *reinterpret_cast<float*>(row + 12) = rstd;
                }
            }
        }
    }
}

__global__ __launch_bounds__(512) void retr_stats_tight_kernel(StatsTArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // waves w and w + 4 share a SIMD: row blocks (j, 7 - j) there, 18 k-steps per SIMD and tile whatever j
    switch (w) {
        case 0: stats_tight_role<0>(a, smem, w, lane); break;
        case 1: stats_tight_role<1>(a, smem, w, lane); break;
        case 2: stats_tight_role<2>(a, smem, w, lane); break;
        case 3: stats_tight_role<3>(a, smem, w, lane); break;
        case 4: stats_tight_role<7>(a, smem, w, lane); break;
        case 5: stats_tight_role<6>(a, smem, w, lane); break;
        case 6: stats_tight_role<5>(a, smem, w, lane); break;
        default: stats_tight_role<4>(a, smem, w, lane); break;
    }
}

}  // namespace svps

extern "C" int svps_retr_stats_tight_fwd(const void* feat, const float* ty, const float* tx, const void* rk_hi, const void* rk_lo,
                                         const float* rbk, float lnk_eps, const void* rv_hi, const void* rv_lo, const float* rbv,
                                         float lnv_eps, void* aux, int T, int H, int W, int D, int flags, void* stream_) {
    if (!feat || !rk_hi || !rk_lo || !rbk || !rv_hi || !rv_lo || !rbv || !aux || ((ty == nullptr) != (tx == nullptr))) return SVPS_ERR_BAD_ARG;
    if (D != svps::kD || T <= 0 || H <= 0 || W <= 0) return SVPS_ERR_BAD_SHAPE;
    if ((size_t)H * W > svps::kMaxFramePixels) return SVPS_ERR_BAD_SHAPE;
    const int HW = H * W;
    const int tiles = (HW + svps::kTilePx - 1) / svps::kTilePx;
    int chunks = svps_pick_chunks(T, tiles, svps_num_cus());
    const int tpw = (tiles + chunks - 1) / chunks;
    chunks = (tiles + tpw - 1) / tpw;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    svps_prof_mark(SVPS_KERNEL_RETR_STATS, 0, stream);
    const svps::StatsTArgs ak{static_cast<const __bf16*>(feat), ty, tx, static_cast<const _Float16*>(rk_hi),
                              static_cast<const _Float16*>(rk_lo), rbk, static_cast<__bf16*>(aux), lnk_eps, HW, H, W, tpw, 0, (flags & SVPS_FLAG_MAP_F16) ? 1 : 0};
    hipLaunchKernelGGL(svps::retr_stats_tight_kernel, dim3(chunks, T), dim3(512), svps::StatsTLds::total, stream, ak);
    const svps::StatsTArgs av{static_cast<const __bf16*>(feat), nullptr, nullptr, static_cast<const _Float16*>(rv_hi),
                              static_cast<const _Float16*>(rv_lo), rbv, static_cast<__bf16*>(aux), lnv_eps, HW, H, W, tpw, 1, (flags & SVPS_FLAG_MAP_F16) ? 1 : 0};
    hipLaunchKernelGGL(svps::retr_stats_tight_kernel, dim3(chunks, T), dim3(512), svps::StatsTLds::total, stream, av);
    svps_prof_mark(SVPS_KERNEL_RETR_STATS, 1, stream);
    return (int)hipGetLastError();
}
