// K6 - full-resolution panoptic post-process on the GPU.
//
// The reference materialises the K kept slot masks at full resolution (K x 8 MB fp32 at 1024x2048),
// copies them to the host and runs NumPy loops over them (PostProcessPanopticInstances.mask_removal /
// get_ids_area, mmdet/models/detectors/vps_temporal_slots.py:564-657, :724-757, and the argmax + relabel
// of simple_test, :411-435). Here the bilinear x4 upsampling (:697-698) is fused into every consumer and
// recomputed from the low-resolution logits, and the order-dependent "first come, first served" pixel
// claiming of mask_removal is reduced to small integer tables:
//
//   * softmax probabilities over the K kept slots sum to 1, so at most TWO slots can reach the 0.4 pixel
//     threshold at a pixel. pp_candidates_kernel stores, per pixel, the (<= 2) thing slots that do
//     (in score order), and counts  n[i] = |{p : i candidate}|  and  N[i][j] = |{p : i and j candidates}|.
//   * the keep / drop decision of thing i (processed by descending score) needs
//         overlap_i = |{p : i candidate and p already claimed by a kept thing of the same class}|
//     and the only other candidate of such a pixel is that thing, so overlap_i = sum_j N[j][i] over kept,
//     earlier, same-class j: the sequential part runs on the K x K table on the host (no pixel work).
//   * pp_argmax_kernel then evaluates, per pixel, the masks AFTER removal - a stuff slot keeps its
//     upsampled logit, a kept thing keeps it only where it is the first kept candidate and is 0 elsewhere
//     (zeros take part in the argmax, like in the reference) - and takes the first-max argmax in a given
//     slot order, applies a lookup table (stuff de-duplication / final id relabel) and histograms the ids
//     (segment areas). The small-area filter loop and the relabel reuse it with other slot lists.
//
// All float arithmetic mirrors the reference operation by operation (this file is compiled with
// -ffp-contract=off: a fused multiply-add would round differently from torch / NumPy), so the
// upsampled logits are bit-identical to the CPU oracle and integer outputs can only differ where a
// softmax probability sits within rounding of the 0.4 threshold.
#include <hip/hip_runtime.h>

#include "../../include/slotvps_hip.h"

namespace svps {

struct Taps {
    int o00, o01, o10, o11;      // offsets of the four taps inside one [h, w] mask
    float hy0, hy1, wx0, wx1;
};

__device__ __forceinline__ void axis_taps(int dst, int n_out, int n_in, int& i0, int& i1, float& l0, float& l1) {
    const float scale = (float)n_in / (float)n_out;
    float src = scale * ((float)dst + 0.5f) - 0.5f;          // area_pixel_compute_source_index, align_corners=False
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    i0 = i0 < n_in - 1 ? i0 : n_in - 1;
    i1 = i0 + 1 < n_in - 1 ? i0 + 1 : n_in - 1;
    l1 = src - (float)i0;
    l0 = 1.f - l1;
}

__device__ __forceinline__ Taps make_taps(int Y, int X, int h, int w, int H, int W) {
    int y0, y1, x0, x1;
    Taps t;
    axis_taps(Y, H, h, y0, y1, t.hy0, t.hy1);
    axis_taps(X, W, w, x0, x1, t.wx0, t.wx1);
    t.o00 = y0 * w + x0; t.o01 = y0 * w + x1; t.o10 = y1 * w + x0; t.o11 = y1 * w + x1;
    return t;
}

// upsample_bilinear2d: h0 * (w0 * a + w1 * b) + h1 * (w0 * c + w1 * d)
__device__ __forceinline__ float upsample(const float* __restrict__ m, const Taps& t) {
    const float top = m[t.o00] * t.wx0 + m[t.o01] * t.wx1;
    const float bot = m[t.o10] * t.wx0 + m[t.o11] * t.wx1;
    return t.hy0 * top + t.hy1 * bot;
}

// masks [K, h, w]: low-resolution logits of the kept slots in DESCENDING SCORE order; is_thing [K].
// cand [H*W, 2] (255 = none), counts [K], pairs [K, K] (pairs[i*K + j], i < j) - zeroed by the caller.
// grid = (ceil(W / 256), H): one output row per blockIdx.y.
__global__ __launch_bounds__(256) void pp_candidates_kernel(const float* __restrict__ masks,
                                                            const uint8_t* __restrict__ is_thing, int K, int h, int w,
                                                            int H, int W, float thr, uint8_t* __restrict__ cand,
                                                            int* __restrict__ counts, int* __restrict__ pairs) {
    __shared__ int lcount[256];
    lcount[threadIdx.x] = 0;
    __syncthreads();
    const int Y = blockIdx.y, X = blockIdx.x * 256 + threadIdx.x;
    int c0 = 255, c1 = 255;
    if (X < W) {
        const Taps t = make_taps(Y, X, h, w, H, W);
        const int hw = h * w;
        float mx = -INFINITY;
        const float* m = masks;
        for (int k = 0; k < K; ++k, m += hw) mx = fmaxf(mx, upsample(m, t));
        // softmax over the K slots exactly as the reference evaluates it: e_k = exp(u_k - max), p_k = e_k / sum.
        // Only the two largest thing terms can reach a threshold > 1/3, so they are tracked in the same pass.
        float sum = 0.f, e0 = -1.f, e1 = -1.f;
        int k0 = 255, k1 = 255;
        m = masks;
        for (int k = 0; k < K; ++k, m += hw) {
            const float e = expf(upsample(m, t) - mx);
            sum += e;
            if (is_thing[k]) {
                if (e > e0) { e1 = e0; k1 = k0; e0 = e; k0 = k; }
                else if (e > e1) { e1 = e; k1 = k; }
            }
        }
        const bool p0 = k0 != 255 && e0 / sum >= thr, p1 = k1 != 255 && e1 / sum >= thr;
        const int a = p0 ? k0 : 255, b = p1 ? k1 : 255;
        c0 = a < b ? a : b;                              // score order = index order
        c1 = a < b ? b : a;
        if (c0 == 255) c1 = 255;
        const size_t px = (size_t)Y * W + X;
        cand[2 * px] = (uint8_t)c0;
        cand[2 * px + 1] = (uint8_t)c1;
        if (c0 != 255) atomicAdd(&lcount[c0], 1);
        if (c1 != 255) atomicAdd(&lcount[c1], 1);
    }
    // pair counts: wave-aggregated - lanes holding the same (c0, c1) pair elect one atomicAdd
    {
        int key = (c1 != 255) ? (c0 << 8 | c1) : -1;
        unsigned long long todo = __ballot(key >= 0);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int lk = __shfl(key, leader);
            const unsigned long long same = __ballot(key == lk);
            if ((int)(threadIdx.x & 63) == leader) atomicAdd(&pairs[(lk >> 8) * K + (lk & 255)], __popcll(same));
            todo &= ~same;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < K && lcount[threadIdx.x]) atomicAdd(&counts[threadIdx.x], lcount[threadIdx.x]);
}

// sel [n]: slots (indices into the sorted kept list) in the order the argmax runs over; sel_thing [n];
// kept [K]: 1 if the slot survived mask_removal (claimer test); lut [n]: id written for position j.
// out_ids [H*W] uint8 (nullable), hist [256] (nullable, zeroed by the caller), out_masks [n, H*W] (nullable).
__global__ __launch_bounds__(256) void pp_argmax_kernel(const float* __restrict__ masks, const uint8_t* __restrict__ sel,
                                                        const uint8_t* __restrict__ sel_thing, int n,
                                                        const uint8_t* __restrict__ kept, const uint8_t* __restrict__ cand,
                                                        const uint8_t* __restrict__ lut, int h, int w, int H, int W,
                                                        uint8_t* __restrict__ out_ids, int* __restrict__ hist,
                                                        float* __restrict__ out_masks) {
    __shared__ int lhist[256];
    __shared__ uint8_t s_sel[256], s_thing[256], s_lut[256];
    lhist[threadIdx.x] = 0;
    if ((int)threadIdx.x < n) {
        s_sel[threadIdx.x] = sel[threadIdx.x];
        s_thing[threadIdx.x] = sel_thing[threadIdx.x];
        s_lut[threadIdx.x] = lut[threadIdx.x];
    }
    __syncthreads();
    const int Y = blockIdx.y, X = blockIdx.x * 256 + threadIdx.x;
    if (X < W) {
        const Taps t = make_taps(Y, X, h, w, H, W);
        const int hw = h * w;
        const size_t px = (size_t)Y * W + X;
        const int c0 = cand[2 * px], c1 = cand[2 * px + 1];
        int claimer = 255;
        if (c0 != 255 && kept[c0]) claimer = c0;
        else if (c1 != 255 && kept[c1]) claimer = c1;
        float best = -INFINITY;
        int bj = 0;
        for (int j = 0; j < n; ++j) {
            const int k = s_sel[j];
            float v = 0.f;
            if (!s_thing[j] || k == claimer) v = upsample(masks + k * hw, t);
            if (out_masks) out_masks[(size_t)j * H * W + px] = v;
            if (v > best) { best = v; bj = j; }      // first maximum wins (torch.argmax)
        }
        const int id = n > 0 ? s_lut[bj] : 0;
        if (out_ids) out_ids[px] = (uint8_t)id;
        if (hist) atomicAdd(&lhist[id], 1);
    }
    __syncthreads();
    if (hist && lhist[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lhist[threadIdx.x]);
}

}  // namespace svps

extern "C" int svps_panoptic_candidates(const float* masks, const uint8_t* is_thing, int K, int h, int w, int H, int W,
                                        float pixel_threshold, uint8_t* cand, int* counts, int* pairs, void* stream_) {
    if (!masks || !is_thing || !cand || !counts || !pairs) return SVPS_ERR_BAD_ARG;
    if (K <= 0 || K > 255 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return SVPS_ERR_BAD_SHAPE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!(pixel_threshold > 1.f / 3.f)) return SVPS_ERR_BAD_ARG;      // at most two slots may reach the threshold
    svps_prof_mark(SVPS_KERNEL_PANOPTIC_POST, 0, stream);
    hipLaunchKernelGGL(svps::pp_candidates_kernel, dim3((W + 255) / 256, H), dim3(256), 0, stream, masks, is_thing, K, h, w, H, W,
                       pixel_threshold, cand, counts, pairs);
    svps_prof_mark(SVPS_KERNEL_PANOPTIC_POST, 1, stream);
    return (int)hipGetLastError();
}

extern "C" int svps_panoptic_argmax(const float* masks, const uint8_t* sel, const uint8_t* sel_thing, int n,
                                    const uint8_t* kept, const uint8_t* cand, const uint8_t* lut, int h, int w, int H,
                                    int W, uint8_t* out_ids, int* hist, float* out_masks, void* stream_) {
    if (!masks || !cand || !kept || (n > 0 && (!sel || !sel_thing || !lut))) return SVPS_ERR_BAD_ARG;
    if (n < 0 || n > 255 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return SVPS_ERR_BAD_SHAPE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    svps_prof_mark(SVPS_KERNEL_PANOPTIC_POST, 0, stream);
    hipLaunchKernelGGL(svps::pp_argmax_kernel, dim3((W + 255) / 256, H), dim3(256), 0, stream, masks, sel, sel_thing, n, kept, cand,
                       lut, h, w, H, W, out_ids, hist, out_masks);
    svps_prof_mark(SVPS_KERNEL_PANOPTIC_POST, 1, stream);
    return (int)hipGetLastError();
}
