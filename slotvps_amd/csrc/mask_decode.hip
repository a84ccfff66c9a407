// K2 - slot -> mask dot-product decode for gfx950.
//
// Replaces VPS_Temporal_Slots.generate_final_outputs (main branch) of the reference,
// mmdet/models/detectors/vps_temporal_slots.py:144-160:
//
//     g[p, :]  = feat_bn(f)[p, :]            = scale (.) f[p, :] + shift        (eval BatchNorm2d)
//     gh[p, :] = g[p, :] / max(||g[p, :]||_2, 1e-12)                            (F.normalize)
//     m[l, p]  = sum_c gh[p, c] * e[l, c]                                       (einsum)
//     m[l, p]  = fg_scale * m[l, p] + fg_shift                                  (fg_bn, scalar)
//
// Folded so that the MFMA runs on the raw bf16 feature tile exactly as it arrives from HBM:
//     m[l, p] = ( sum_c (e[l,c] * scale[c]) * f[p,c]  +  sum_c e[l,c] * shift[c] ) / max(||g_p||, 1e-12)
// The slot operand e (.) scale is carried as bf16 hi + lo (two MFMAs), so the only rounding left is
// fp32 accumulation; the feature map itself is the bf16 tensor the head stores.
//
// HBM-bound: reads T*HW*512 B, writes T*L*HW*4 B. Layout and staging are those of K1 (32-pixel
// tiles by LDS-DMA into a 4-stage ring, wave w owns slots [32w, 32w+32), pixel = lane), so the
// logits leave the accumulators as 128-B pixel-contiguous row segments of the [T, L, HW] output.
#include "common.h"
#include "../../include/slotvps_hip.h"

namespace svps {

template <int NW, int NST>
struct DecLds {
    static constexpr int ring = 0;                         // NST feature tiles
    static constexpr int affine = NST * kTileBytes;        // scale[256], shift[256] fp32
    static constexpr int norm = affine + 2 * kD * 4;       // inv-norm per tile pixel [32]
    static constexpr int cshift = norm + kTilePx * 4;      // per-slot constant [NW * 32]
    static constexpr int amax = cshift + NW * 32 * 4;      // per-wave argmax candidates [NW][32] float2
    static constexpr int total = amax + NW * kTilePx * 8;
};

template <int NW, int NST, bool ARGMAX, typename OutT>
__global__ __launch_bounds__(NW * 64) void mask_decode_kernel(
    const __bf16* __restrict__ feat,     // [T, HW, 256]
    const float* __restrict__ embed,     // [T, L, 256]
    const float* __restrict__ bn_scale,  // [256]
    const float* __restrict__ bn_shift,  // [256]
    float fg_scale, float fg_shift,
    OutT* __restrict__ out,              // [T, L, HW]
    uint8_t* __restrict__ slot_argmax,   // [T, HW] or null
    int L, int HW, int tiles_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using Lds = DecLds<NW, NST>;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int t = blockIdx.y, c = blockIdx.x;

    const int px_begin = c * tiles_per_chunk * kTilePx;
    int px_end = px_begin + tiles_per_chunk * kTilePx;
    px_end = px_end < HW ? px_end : HW;
    const int nt = (px_end - px_begin + kTilePx - 1) / kTilePx;

    float* aff = reinterpret_cast<float*>(smem + Lds::affine);
    float* inv_norm = reinterpret_cast<float*>(smem + Lds::norm);
    float* cs = reinterpret_cast<float*>(smem + Lds::cshift);
    float2* am = reinterpret_cast<float2*>(smem + Lds::amax);

    for (int i = tid; i < kD; i += NW * 64) {
        aff[i] = bn_scale[i];
        aff[kD + i] = bn_shift[i];
    }
    __syncthreads();

    // ---- slot operand (e * scale) as bf16 hi/lo A fragments; per-slot constant e . shift ------
    bf16x8 eh[16], el[16];
    {
        const int slot = 32 * w + r;
        const float* erow = embed + ((size_t)t * L + (slot < L ? slot : 0)) * kD + 8 * h;
        float dot = 0.f;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(erow + 16 * ks);
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(erow + 16 * ks + 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ch = 16 * ks + 8 * h + j;
                float x = j < 4 ? x0[j] : x1[j - 4];
                if (slot >= L) x = 0.f;
                dot += x * aff[kD + ch];
                const float xs = x * aff[ch];
                const __bf16 hi = (__bf16)xs;
                eh[ks][j] = hi;
                el[ks][j] = (__bf16)(xs - (float)hi);
            }
        }
        dot = wave_half_xor_sum(dot);
        if (h == 0) cs[32 * w + r] = dot;
    }
    wait_vm<0>();
    __syncthreads();
    float csr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) csr[i] = cs[32 * w + acc_row(i, h)];

    const char* fb = reinterpret_cast<const char*>(feat) + (size_t)t * HW * kRowBytes;
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nt) dma_tile<NW>(fb, px_begin + s * kTilePx, HW - 1, smem + Lds::ring + s * kTileBytes, w, lane);

    constexpr int PIECES = kTilePx / 2 / NW;
    // thread -> (pixel, channel octet set) for the norm pass: 8 * NW / 4 threads per pixel
    constexpr int TPP = NW * 64 / kTilePx;       // threads per pixel (8 or 16)
    constexpr int CPT = 32 / TPP;                // 16-B chunks per thread (4 or 2)
    const int npx = tid / TPP, nsub = tid % TPP;

    for (int it = 0; it < nt; ++it) {
        if constexpr (NST == 4) {
            if (it + 2 < nt) wait_vm<2 * PIECES>();
            else if (it + 1 < nt) wait_vm<PIECES>();
            else wait_vm<0>();
        } else {
            wait_vm<0>();
        }
        wg_barrier();
        if (it + NST - 1 < nt)
            dma_tile<NW>(fb, px_begin + (it + NST - 1) * kTilePx, HW - 1,
                         smem + Lds::ring + ((it + NST - 1) % NST) * kTileBytes, w, lane);
        const char* ft = smem + Lds::ring + (it % NST) * kTileBytes;

        // -- ||scale * f + shift||^2 per pixel -------------------------------------------------
        {
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                const int chunk = nsub + TPP * i;
                const bf16x8 x = *reinterpret_cast<const bf16x8*>(ft + npx * kRowBytes + ((chunk ^ swz(npx)) * 16));
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float g = (float)x[j] * aff[8 * chunk + j] + aff[kD + 8 * chunk + j];
                    ss += g * g;
                }
            }
#pragma unroll
            for (int m = 1; m < TPP; m <<= 1) ss += __shfl_xor(ss, m);
            if (nsub == 0) inv_norm[npx] = 1.f / fmaxf(sqrtf(ss), 1e-12f);
        }

        // -- (e * scale) . f for 32 slots x 32 pixels --------------------------------------------
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const bf16x8 ff = read_row_frag(ft, ks, r, h);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(el[ks], ff, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(eh[ks], ff, s, 0, 0, 0);
        }
        wg_barrier();  // inv_norm of this tile visible

        const int px = px_begin + it * kTilePx + r;
        const float inr = inv_norm[r];
        const bool pix_ok = px < px_end;
        float best = -INFINITY;
        int best_slot = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int slot = 32 * w + acc_row(i, h);
            const float m = (s[i] + csr[i]) * inr * fg_scale + fg_shift;
            if (slot < L) {
                if (pix_ok) out[((size_t)t * L + slot) * HW + px] = (OutT)m;
                if constexpr (ARGMAX) {
                    if (m > best) { best = m; best_slot = slot; }  // slots ascend with i within a lane
                }
            }
        }
        if constexpr (ARGMAX) {
            // lanes r and r+32 hold interleaved slot groups of the same pixel
            const float ob = __shfl_xor(best, 32);
            const int os = __shfl_xor(best_slot, 32);
            if (ob > best || (ob == best && os < best_slot)) { best = ob; best_slot = os; }
            if (h == 0) am[w * kTilePx + r] = make_float2(best, __int_as_float(best_slot));
            wg_barrier();
            if (w == 0 && h == 0 && pix_ok) {
                float b = -INFINITY;
                int bs = 0x7fffffff;
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) {
                    const float2 cnd = am[ww * kTilePx + r];
                    const int sl = __float_as_int(cnd.y);
                    if (cnd.x > b || (cnd.x == b && sl < bs)) { b = cnd.x; bs = sl; }
                }
                slot_argmax[(size_t)t * HW + px] = (uint8_t)bs;
            }
        }
    }
}

}  // namespace svps

namespace {

int dec_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return n;
}

template <int NW, int NST, bool ARGMAX, typename OutT>
hipError_t launch_decode(const void* feat, const float* embed, const float* bn_scale, const float* bn_shift,
                         float fg_scale, float fg_shift, void* out, uint8_t* slot_argmax, int T, int L,
                         int HW, hipStream_t stream) {
    using Lds = svps::DecLds<NW, NST>;
    auto kern = svps::mask_decode_kernel<NW, NST, ARGMAX, OutT>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, Lds::total);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    // two co-resident workgroups per CU (LDS 2 x ~70 KiB): one computes while the other waits
    const int tiles = (HW + svps::kTilePx - 1) / svps::kTilePx;
    int chunks = 2 * dec_num_cus() / T;
    if (chunks < 1) chunks = 1;
    if (chunks > tiles) chunks = tiles;
    const int tpc = (tiles + chunks - 1) / chunks;
    chunks = (tiles + tpc - 1) / tpc;
    hipLaunchKernelGGL(kern, dim3(chunks, T), dim3(NW * 64), Lds::total, stream,
                       static_cast<const __bf16*>(feat), embed, bn_scale, bn_shift, fg_scale, fg_shift,
                       static_cast<OutT*>(out), slot_argmax, L, HW, tpc);
    return hipGetLastError();
}

template <int NW, int NST>
hipError_t dispatch_decode(const void* feat, const float* embed, const float* bn_scale, const float* bn_shift,
                           float fg_scale, float fg_shift, void* out, uint8_t* slot_argmax, int T, int L,
                           int HW, int flags, hipStream_t stream) {
    const bool bf = flags & SVPS_FLAG_OUT_BF16;
    if (slot_argmax)
        return bf ? launch_decode<NW, NST, true, __bf16>(feat, embed, bn_scale, bn_shift, fg_scale, fg_shift, out, slot_argmax, T, L, HW, stream)
                  : launch_decode<NW, NST, true, float>(feat, embed, bn_scale, bn_shift, fg_scale, fg_shift, out, slot_argmax, T, L, HW, stream);
    return bf ? launch_decode<NW, NST, false, __bf16>(feat, embed, bn_scale, bn_shift, fg_scale, fg_shift, out, slot_argmax, T, L, HW, stream)
              : launch_decode<NW, NST, false, float>(feat, embed, bn_scale, bn_shift, fg_scale, fg_shift, out, slot_argmax, T, L, HW, stream);
}

}  // namespace

extern "C" int svps_mask_decode_fwd(const void* feat, const float* embed, const float* bn_scale,
                                    const float* bn_shift, float fg_scale, float fg_shift, void* out,
                                    uint8_t* slot_argmax, int T, int L, int HW, int D, int flags,
                                    void* stream_) {
    if (!feat || !embed || !bn_scale || !bn_shift || !out) return SVPS_ERR_BAD_ARG;
    if (D != svps::kD || T <= 0 || L <= 0 || L > 256 || HW <= 0) return SVPS_ERR_BAD_SHAPE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    svps_prof_mark(SVPS_KERNEL_MASK_DECODE, 0, stream);
    hipError_t e = L <= 128 ? dispatch_decode<4, 4>(feat, embed, bn_scale, bn_shift, fg_scale, fg_shift, out,
                                                    slot_argmax, T, L, HW, flags, stream)
                            : dispatch_decode<8, 4>(feat, embed, bn_scale, bn_shift, fg_scale, fg_shift, out,
                                                    slot_argmax, T, L, HW, flags, stream);
    svps_prof_mark(SVPS_KERNEL_MASK_DECODE, 1, stream);
    return (int)e;
}
