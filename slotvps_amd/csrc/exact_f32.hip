// Exact mode: the pixel side of the slot head with fp32 STORAGE and fp32 arithmetic, for gfx950.
//
// The reference runs this path in fp32 (fp16_enabled = False, mmdet/models/detectors/vps_temporal_slots.py:55).
// The fast path of this library stores the fused level maps (and, in its first form, k / v) as bf16; these
// kernels store nothing below fp32 and use only fp32 FMA, so that the whole head can be compared FREE-RUNNING
// with the reference's own fp32 outputs at the tolerance the reference itself reproduces to (summation order).
// They are written for that purpose - straightforward LDS tiling on the vector ALU, one rounding per operation
// (the fp32 matrix instructions of gfx950 run at the vector rate, so they would buy nothing here) - and run one
// to two orders of magnitude slower than the bf16 kernels. Same C-ABI conventions, same pixel-major
// layouts, `float` instead of bf16.
//
//   svps_level_fuse_f32_fwd    MultiScaleDynamicMaskHead.forward feature side   dynamic_mask_head.py:171-188
//   svps_kv_project_f32_fwd    k = norm_k(to_k(f + pos)), v = norm_v(to_v(f))    dynamic_mask_head.py:428-433
//   svps_slot_attn_f32_fwd     logits, softmax over slots, attn.v, norm1, ReLU   dynamic_mask_head.py:435-459
//   svps_mask_decode_f32_fwd   generate_final_outputs                            vps_temporal_slots.py:144-160
#include "common.h"
#include "../../include/slotvps_hip.h"

namespace svps {
namespace exact {

constexpr int kPxTile = 16;        // pixels per workgroup step
constexpr int kThreads = 256;      // one thread per output channel / per slot

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m);
    return x;
}
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x = fmaxf(x, __shfl_xor(x, m));
    return x;
}

// torch upsample_bilinear2d, align_corners = False, scale 2: source index and weight of output index `d` on an axis of
// `n` source samples (aten/src/ATen/native/UpSample.h area_pixel_compute_source_index)
__device__ __forceinline__ void tap2x(int d, int n, int& i0, int& i1, float& lam) {
    float src = (d + 0.5f) * 0.5f - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    i1 = i0 + (i0 < n - 1 ? 1 : 0);
    lam = src - (float)i0;
}

// ---- level fusion -----------------------------------------------------------------------------------------------
// out[p, :] = Wc . cat(bilinear_x2(prev)[p, 0:256], cur[p, 0:128]) + bc   (level 0: cat(cur, cur, cur), :183)
// thread = output channel, workgroup step = kPxTile pixels whose 384-vector is staged in LDS.
__global__ __launch_bounds__(kThreads) void level_fuse_f32_kernel(const float* __restrict__ cur,    // [T, 128, H, W]
                                                                  const float* __restrict__ prev,   // [T, (H/2)(W/2), 256] or null
                                                                  const float* __restrict__ wT,     // [384, 256]
                                                                  const float* __restrict__ bc,     // [256]
                                                                  float* __restrict__ out,          // [T, H*W, 256]
                                                                  int H, int W) {
    __shared__ float cat[kPxTile][384];
    const int t = blockIdx.y, c = threadIdx.x;
    const int HW = H * W;
    const int p0 = blockIdx.x * kPxTile;
    const float* curt = cur + (size_t)t * 128 * HW;
    for (int i = c; i < kPxTile * 128; i += kThreads) {            // incoming 128-channel map: channel-major source
        const int ch = i / kPxTile, px = i % kPxTile;
        const int p = p0 + px;
        const float v = p < HW ? curt[(size_t)ch * HW + p] : 0.f;
        if (prev) cat[px][256 + ch] = v;
        else { cat[px][ch] = v; cat[px][128 + ch] = v; cat[px][256 + ch] = v; }
    }
    if (prev) {
        const int Hs = H / 2, Ws = W / 2;
        const float* pt = prev + (size_t)t * Hs * Ws * 256;
        for (int px = 0; px < kPxTile; ++px) {
            const int p = p0 + px;
            float v = 0.f;
            if (p < HW) {
                const int y = p / W, x = p - y * W;
                int y0, y1, x0, x1;
                float ly, lx;
                tap2x(y, Hs, y0, y1, ly);
                tap2x(x, Ws, x0, x1, lx);
                const float a = pt[((size_t)y0 * Ws + x0) * 256 + c], b = pt[((size_t)y0 * Ws + x1) * 256 + c];
                const float d = pt[((size_t)y1 * Ws + x0) * 256 + c], e = pt[((size_t)y1 * Ws + x1) * 256 + c];
                v = (1.f - ly) * ((1.f - lx) * a + lx * b) + ly * ((1.f - lx) * d + lx * e);
            }
            cat[px][c] = v;
        }
    }
    __syncthreads();
    float acc[kPxTile];
    const float b = bc[c];
#pragma unroll
    for (int px = 0; px < kPxTile; ++px) acc[px] = b;
    for (int k = 0; k < 384; ++k) {
        const float w = wT[k * 256 + c];
#pragma unroll
        for (int px = 0; px < kPxTile; ++px) acc[px] = fmaf(w, cat[px][k], acc[px]);
    }
#pragma unroll
    for (int px = 0; px < kPxTile; ++px)
        if (p0 + px < HW) out[((size_t)t * HW + p0 + px) * 256 + c] = acc[px];
}

// ---- k / v projection + LayerNorm -------------------------------------------------------------------------------
// One projection per launch (blockIdx.z: 0 = key with position embedding, 1 = value). Two-pass LayerNorm statistics
// (mean, then centred sum of squares), biased variance, eps inside the square root - torch.nn.LayerNorm.
__global__ __launch_bounds__(kThreads) void kv_project_f32_kernel(
    const float* __restrict__ feat,                                  // [T, HW, 256]
    const float* __restrict__ pos_y, const float* __restrict__ pos_x,   // [H, 128], [W, 128] or null
    const float* __restrict__ wkT, const float* __restrict__ bk, const float* __restrict__ gk, const float* __restrict__ ek, float eps_k,
    const float* __restrict__ wvT, const float* __restrict__ bv, const float* __restrict__ gv, const float* __restrict__ ev, float eps_v,
    float* __restrict__ k_out, float* __restrict__ v_out, int HW, int W) {
    __shared__ float x[kPxTile][256];
    __shared__ float u[kPxTile][256 + 1];
    const int t = blockIdx.y, c = threadIdx.x, proj = blockIdx.z;
    const int p0 = blockIdx.x * kPxTile;
    const float* ft = feat + (size_t)t * HW * 256;
    for (int px = 0; px < kPxTile; ++px) {
        const int p = p0 + px;
        float v = p < HW ? ft[(size_t)p * 256 + c] : 0.f;
        if (proj == 0 && pos_y && p < HW) {
            const int yy = p / W, xx = p - yy * W;
            v += c < 128 ? pos_y[yy * 128 + c] : pos_x[xx * 128 + (c - 128)];     // with_pos_embed, :575-583
        }
        x[px][c] = v;
    }
    __syncthreads();
    const float* wT = proj ? wvT : wkT;
    float acc[kPxTile];
    const float b = (proj ? bv : bk)[c];
#pragma unroll
    for (int px = 0; px < kPxTile; ++px) acc[px] = b;
    for (int k = 0; k < 256; ++k) {
        const float w = wT[k * 256 + c];
#pragma unroll
        for (int px = 0; px < kPxTile; ++px) acc[px] = fmaf(w, x[px][k], acc[px]);
    }
#pragma unroll
    for (int px = 0; px < kPxTile; ++px) u[px][c] = acc[px];
    __syncthreads();
    // wave w normalises pixels 4w .. 4w + 3: lane holds 4 channels of a pixel row
    const int w = c >> 6, lane = c & 63;
    const float g0 = (proj ? gv : gk)[lane], g1 = (proj ? gv : gk)[lane + 64], g2 = (proj ? gv : gk)[lane + 128], g3 = (proj ? gv : gk)[lane + 192];
    const float e0 = (proj ? ev : ek)[lane], e1 = (proj ? ev : ek)[lane + 64], e2 = (proj ? ev : ek)[lane + 128], e3 = (proj ? ev : ek)[lane + 192];
    const float eps = proj ? eps_v : eps_k;
    float* ot = (proj ? v_out : k_out) + (size_t)t * HW * 256;
    for (int j = 0; j < kPxTile / 4; ++j) {
        const int px = (kPxTile / 4) * w + j;
        const float a0 = u[px][lane], a1 = u[px][lane + 64], a2 = u[px][lane + 128], a3 = u[px][lane + 192];
        const float mean = wave_sum((a0 + a1) + (a2 + a3)) * (1.f / 256.f);
        const float d0 = a0 - mean, d1 = a1 - mean, d2 = a2 - mean, d3 = a3 - mean;
        const float var = wave_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) * (1.f / 256.f);
        const float rstd = 1.f / sqrtf(var + eps);
        if (p0 + px < HW) {
            float* o = ot + (size_t)(p0 + px) * 256;
            o[lane] = d0 * rstd * g0 + e0;
            o[lane + 64] = d1 * rstd * g1 + e1;
            o[lane + 128] = d2 * rstd * g2 + e2;
            o[lane + 192] = d3 * rstd * g3 + e3;
        }
    }
}

// ---- retriever -----------------------------------------------------------------------------------------------------
// Per workgroup: a contiguous pixel range of one frame, 16 pixels per step.
//   phase 1  thread l: logits[l][px] = q[l, :] . k[px, :]                 (k tile in LDS, q row streamed from L2)
//   phase 2  wave w: softmax over the L slots of pixels 4w .. 4w + 3       (max, exp, sum, divide - F.softmax)
//   phase 3  thread c: o[l][c] += P[l][px] * v[px][c] for all l            (LMAX accumulators in registers)
// Partials [T, C, L, 256] go through slot_attn_finish (fixed-order sum + LayerNorm + ReLU) of slot_attn.hip.
template <int LMAX>
__global__ __launch_bounds__(kThreads) void slot_attn_f32_kernel(const float* __restrict__ q,   // [T, L, 256]
                                                                 const float* __restrict__ k,   // [T, HW, 256]
                                                                 const float* __restrict__ v,
                                                                 float* __restrict__ partial,   // [T, C, L, 256]
                                                                 int L, int HW, int px_per_chunk) {
    __shared__ float kt[kPxTile][256];
    __shared__ float vt[kPxTile][256];
    __shared__ __attribute__((aligned(16))) float P[kPxTile][LMAX];
    const int t = blockIdx.y, ck = blockIdx.x, C = gridDim.x, tid = threadIdx.x;
    const int w = tid >> 6, lane = tid & 63;
    const int pb = ck * px_per_chunk;
    int pe = pb + px_per_chunk;
    pe = pe < HW ? pe : HW;
    float o[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) o[l] = 0.f;
    const float* kf = k + (size_t)t * HW * 256;
    const float* vf = v + (size_t)t * HW * 256;
    for (int p0 = pb; p0 < pe; p0 += kPxTile) {
        __syncthreads();                                           // previous step done with kt / vt / P
        for (int px = 0; px < kPxTile; ++px) {
            const int p = p0 + px;
            kt[px][tid] = p < pe ? kf[(size_t)p * 256 + tid] : 0.f;
            vt[px][tid] = p < pe ? vf[(size_t)p * 256 + tid] : 0.f;
        }
        __syncthreads();
        for (int l = tid; l < LMAX; l += kThreads) {
            float acc[kPxTile];
#pragma unroll
            for (int px = 0; px < kPxTile; ++px) acc[px] = 0.f;
            if (l < L) {
                const float4* qr = reinterpret_cast<const float4*>(q + ((size_t)t * L + l) * 256);
                for (int c4 = 0; c4 < 64; ++c4) {
                    const float4 qq = qr[c4];
#pragma unroll
                    for (int px = 0; px < kPxTile; ++px) {
                        acc[px] = fmaf(qq.x, kt[px][4 * c4], acc[px]);
                        acc[px] = fmaf(qq.y, kt[px][4 * c4 + 1], acc[px]);
                        acc[px] = fmaf(qq.z, kt[px][4 * c4 + 2], acc[px]);
                        acc[px] = fmaf(qq.w, kt[px][4 * c4 + 3], acc[px]);
                    }
                }
            }
#pragma unroll
            for (int px = 0; px < kPxTile; ++px) P[px][l] = l < L ? acc[px] : -INFINITY;
        }
        __syncthreads();
        for (int j = 0; j < kPxTile / 4; ++j) {
            const int px = (kPxTile / 4) * w + j;
            float m = -INFINITY;
            for (int l = lane; l < LMAX; l += 64) m = fmaxf(m, P[px][l]);
            m = wave_max(m);
            float s = 0.f;
            for (int l = lane; l < LMAX; l += 64) {
                const float e = l < L ? expf(P[px][l] - m) : 0.f;
                P[px][l] = e;
                s += e;
            }
            s = wave_sum(s);
            const bool live = p0 + px < pe;
            for (int l = lane; l < LMAX; l += 64) P[px][l] = live ? P[px][l] / s : 0.f;
        }
        __syncthreads();
#pragma unroll 1
        for (int px = 0; px < kPxTile; ++px) {
            const float vv = vt[px][tid];
#pragma unroll
            for (int l4 = 0; l4 < LMAX / 4; ++l4) {
                const float4 pp = *reinterpret_cast<const float4*>(&P[px][4 * l4]);      // broadcast read
                o[4 * l4] = fmaf(pp.x, vv, o[4 * l4]);
                o[4 * l4 + 1] = fmaf(pp.y, vv, o[4 * l4 + 1]);
                o[4 * l4 + 2] = fmaf(pp.z, vv, o[4 * l4 + 2]);
                o[4 * l4 + 3] = fmaf(pp.w, vv, o[4 * l4 + 3]);
            }
        }
    }
    float* dst = partial + ((size_t)t * C + ck) * L * 256;
#pragma unroll
    for (int l = 0; l < LMAX; ++l)
        if (l < L) dst[(size_t)l * 256 + tid] = o[l];
}

// Sum of the per-workgroup partials in chunk order + LayerNorm + ReLU (same semantics as slot_attn_finish, two-pass
// statistics); one workgroup per (slot, frame), thread = channel.
__global__ __launch_bounds__(256) void slot_attn_f32_finish(const float* __restrict__ partial, const float* __restrict__ ln_w,
                                                            const float* __restrict__ ln_b, float eps,
                                                            float* __restrict__ out, float* __restrict__ out_pre, int L, int C) {
    __shared__ float red[4];
    const int l = blockIdx.x, t = blockIdx.y, d = threadIdx.x;
    const float* src = partial + ((size_t)t * C * L + l) * 256 + d;
    float acc = 0.f;
    for (int c = 0; c < C; ++c) acc += src[(size_t)c * L * 256];
    if (out_pre) out_pre[((size_t)t * L + l) * 256 + d] = acc;
    auto block_sum = [&](float x) {
        x = wave_sum(x);
        __syncthreads();
        if ((d & 63) == 0) red[d >> 6] = x;
        __syncthreads();
        return (red[0] + red[1]) + (red[2] + red[3]);
    };
    const float mean = block_sum(acc) * (1.f / 256.f);
    const float dev = acc - mean;
    const float var = block_sum(dev * dev) * (1.f / 256.f);
    const float y = dev * (1.f / sqrtf(var + eps)) * ln_w[d] + ln_b[d];
    out[((size_t)t * L + l) * 256 + d] = y > 0.f ? y : 0.f;
}

// ---- mask decode ---------------------------------------------------------------------------------------------------
// m[l, p] = (embed[l, :] . normalize2(feat[p, :] * bn_scale + bn_shift)) * fg_scale + fg_shift
__global__ __launch_bounds__(kThreads) void mask_decode_f32_kernel(const float* __restrict__ feat,     // [T, HW, 256]
                                                                   const float* __restrict__ embed,    // [T, L, 256]
                                                                   const float* __restrict__ bn_scale, const float* __restrict__ bn_shift,
                                                                   float fg_scale, float fg_shift,
                                                                   float* __restrict__ out,            // [T, L, HW]
                                                                   int L, int HW) {
    __shared__ float g[kPxTile][256];
    const int t = blockIdx.y, tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int p0 = blockIdx.x * kPxTile;
    const float* ft = feat + (size_t)t * HW * 256;
    const float sc = bn_scale[tid], sh = bn_shift[tid];
    for (int px = 0; px < kPxTile; ++px) {
        const int p = p0 + px;
        g[px][tid] = p < HW ? ft[(size_t)p * 256 + tid] * sc + sh : 0.f;               // feat_bn (eval), :146
    }
    __syncthreads();
    for (int j = 0; j < kPxTile / 4; ++j) {                                             // F.normalize(p=2, dim=1, eps=1e-12), :147
        const int px = (kPxTile / 4) * w + j;
        const float a0 = g[px][lane], a1 = g[px][lane + 64], a2 = g[px][lane + 128], a3 = g[px][lane + 192];
        const float nrm = fmaxf(sqrtf(wave_sum((a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3))), 1e-12f);
        g[px][lane] = a0 / nrm;
        g[px][lane + 64] = a1 / nrm;
        g[px][lane + 128] = a2 / nrm;
        g[px][lane + 192] = a3 / nrm;
    }
    __syncthreads();
    for (int l = tid; l < L; l += kThreads) {
        float acc[kPxTile];
#pragma unroll
        for (int px = 0; px < kPxTile; ++px) acc[px] = 0.f;
        const float4* er = reinterpret_cast<const float4*>(embed + ((size_t)t * L + l) * 256);
        for (int c4 = 0; c4 < 64; ++c4) {
            const float4 ee = er[c4];
#pragma unroll
            for (int px = 0; px < kPxTile; ++px) {
                acc[px] = fmaf(ee.x, g[px][4 * c4], acc[px]);
                acc[px] = fmaf(ee.y, g[px][4 * c4 + 1], acc[px]);
                acc[px] = fmaf(ee.z, g[px][4 * c4 + 2], acc[px]);
                acc[px] = fmaf(ee.w, g[px][4 * c4 + 3], acc[px]);
            }
        }
        float* o = out + ((size_t)t * L + l) * HW + p0;
#pragma unroll
        for (int px = 0; px < kPxTile; ++px)
            if (p0 + px < HW) o[px] = acc[px] * fg_scale + fg_shift;                     // fg_bn (scalar affine), :153
    }
}

}  // namespace exact
}  // namespace svps

namespace {
int exact_chunks(int T, int HW) {
    const int tile = svps::exact::kPxTile;
    const int tiles = (HW + tile - 1) / tile;
    int chunks = (4 * svps_num_cus() + T - 1) / T;            // ~4 workgroups of 4 waves per CU
    if (chunks < 1) chunks = 1;
    if (chunks > tiles) chunks = tiles;
    return chunks;
}
int exact_px_per_chunk(int HW, int chunks) {
    const int tile = svps::exact::kPxTile;
    const int tiles = (HW + tile - 1) / tile;
    return ((tiles + chunks - 1) / chunks) * tile;
}
}  // namespace

extern "C" int svps_level_fuse_f32_fwd(const float* cur, const float* prev, const float* wT, const float* bc, float* out,
                                       int T, int H, int W, void* stream_) {
    if (!cur || !wT || !bc || !out) return SVPS_ERR_BAD_ARG;
    if (T <= 0 || H <= 0 || W <= 0 || (prev && ((H | W) & 1))) return SVPS_ERR_BAD_SHAPE;
    if ((size_t)H * W > svps::kMaxFramePixels) return SVPS_ERR_BAD_SHAPE;
    const int tiles = (H * W + svps::exact::kPxTile - 1) / svps::exact::kPxTile;
    hipLaunchKernelGGL(svps::exact::level_fuse_f32_kernel, dim3(tiles, T), dim3(svps::exact::kThreads), 0,
                       static_cast<hipStream_t>(stream_), cur, prev, wT, bc, out, H, W);
    return (int)hipGetLastError();
}

extern "C" int svps_kv_project_f32_fwd(const float* feat, const float* pos_y, const float* pos_x, const float* wkT,
                                       const float* bk, const float* lnk_w, const float* lnk_b, float lnk_eps,
                                       const float* wvT, const float* bv, const float* lnv_w, const float* lnv_b,
                                       float lnv_eps, float* k_out, float* v_out, int T, int H, int W, int D, void* stream_) {
    if (!feat || !wkT || !bk || !lnk_w || !lnk_b || !wvT || !bv || !lnv_w || !lnv_b || !k_out || !v_out) return SVPS_ERR_BAD_ARG;
    if ((pos_y == nullptr) != (pos_x == nullptr)) return SVPS_ERR_BAD_ARG;
    if (D != svps::kD || T <= 0 || H <= 0 || W <= 0) return SVPS_ERR_BAD_SHAPE;
    if ((size_t)H * W > svps::kMaxFramePixels) return SVPS_ERR_BAD_SHAPE;
    const int HW = H * W;
    const int tiles = (HW + svps::exact::kPxTile - 1) / svps::exact::kPxTile;
    hipLaunchKernelGGL(svps::exact::kv_project_f32_kernel, dim3(tiles, T, 2), dim3(svps::exact::kThreads), 0,
                       static_cast<hipStream_t>(stream_), feat, pos_y, pos_x, wkT, bk, lnk_w, lnk_b, lnk_eps, wvT, bv,
                       lnv_w, lnv_b, lnv_eps, k_out, v_out, HW, W);
    return (int)hipGetLastError();
}

extern "C" size_t svps_slot_attn_f32_workspace_bytes(int T, int L, int HW) {
    if (T <= 0 || L <= 0 || HW <= 0) return 0;
    int chunks = exact_chunks(T, HW);
    const int ppc = exact_px_per_chunk(HW, chunks);
    chunks = (HW + ppc - 1) / ppc;
    return (size_t)T * chunks * L * svps::kD * sizeof(float);
}

extern "C" int svps_slot_attn_f32_fwd(const float* q, const float* k, const float* v, const float* ln_w, const float* ln_b,
                                      float ln_eps, void* workspace, size_t workspace_bytes, float* out, float* out_pre_ln,
                                      int T, int L, int HW, int D, void* stream_) {
    if (!q || !k || !v || !ln_w || !ln_b || !workspace || !out) return SVPS_ERR_BAD_ARG;
    if (D != svps::kD || T <= 0 || L <= 0 || L > 256 || HW <= 0) return SVPS_ERR_BAD_SHAPE;
    if ((size_t)HW > svps::kMaxFramePixels) return SVPS_ERR_BAD_SHAPE;
    int chunks = exact_chunks(T, HW);
    const int ppc = exact_px_per_chunk(HW, chunks);
    chunks = (HW + ppc - 1) / ppc;
    if (workspace_bytes < (size_t)T * chunks * L * svps::kD * sizeof(float)) return SVPS_ERR_WORKSPACE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    float* partial = static_cast<float*>(workspace);
    if (L <= 128)
        hipLaunchKernelGGL(svps::exact::slot_attn_f32_kernel<128>, dim3(chunks, T), dim3(svps::exact::kThreads), 0, stream,
                           q, k, v, partial, L, HW, ppc);
    else
        hipLaunchKernelGGL(svps::exact::slot_attn_f32_kernel<256>, dim3(chunks, T), dim3(svps::exact::kThreads), 0, stream,
                           q, k, v, partial, L, HW, ppc);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(svps::exact::slot_attn_f32_finish, dim3(L, T), dim3(256), 0, stream, partial, ln_w, ln_b, ln_eps, out,
                       out_pre_ln, L, chunks);
    return (int)hipGetLastError();
}

extern "C" int svps_mask_decode_f32_fwd(const float* feat, const float* embed, const float* bn_scale, const float* bn_shift,
                                        float fg_scale, float fg_shift, float* out, int T, int L, int HW, int D, void* stream_) {
    if (!feat || !embed || !bn_scale || !bn_shift || !out) return SVPS_ERR_BAD_ARG;
    if (D != svps::kD || T <= 0 || L <= 0 || HW <= 0) return SVPS_ERR_BAD_SHAPE;
    const int tiles = (HW + svps::exact::kPxTile - 1) / svps::exact::kPxTile;
    hipLaunchKernelGGL(svps::exact::mask_decode_f32_kernel, dim3(tiles, T), dim3(svps::exact::kThreads), 0,
                       static_cast<hipStream_t>(stream_), feat, embed, bn_scale, bn_shift, fg_scale, fg_shift, out, L, HW);
    return (int)hipGetLastError();
}
