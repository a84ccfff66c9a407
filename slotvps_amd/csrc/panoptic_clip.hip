// K6c - the panoptic post-process of a WHOLE CLIP without the host in its loop.
//
// Same function as panoptic_post.hip (PostProcessPanopticInstances.mask_removal / get_ids_area,
// mmdet/models/detectors/vps_temporal_slots.py:564-657, :724-790, and the argmax + relabel of simple_test :411-435), same
// arithmetic operation by operation (compiled with -ffp-contract=off as well), re-organised twice:
//
//   * the T frames of a clip go through every phase in ONE launch (blockIdx.z = frame), and the order-dependent decisions - the
//     keep / drop loop of mask_removal on the K x K pair table, the stuff de-duplication, the small-area loop, the relabel table - run
//     in single-workgroup kernels on a per-frame STATE block in device memory (layout: SVPS_PPC_* in include/slotvps_hip.h) instead
//     of on the host. The host uploads (K, thing, class) per frame once, enqueues the whole sequence and reads the state back ONCE.
//     The small-area loop has a data-dependent trip count: `rounds` (area pass, step) pairs are enqueued speculatively; a frame
//     that is finished turns the remaining ones into no-ops, a frame that is not is visible in its state (phase < 2) and the host
//     enqueues more rounds (stage 2 | 4) - the decisions are the same either way;
//   * the x4 bilinear upsampling (align_corners = False) is evaluated per 4 x 4 OUTPUT block: output columns 4q .. 4q+3 read the
//     source columns (q-1, q) and (q, q+1) with the weights (0.375, 0.625), (0.125, 0.875), (0.875, 0.125), (0.625, 0.375) - a
//     3 x 3 source patch per slot and thread instead of 16 x 4 taps, the twelve horizontal interpolations shared by the four
//     output rows. Operands and operation order of every output value are those of upsample() in panoptic_post.hip; at the left /
//     top border the reference's tap pair is (0, 1) with weights (1, 0) where this kernel multiplies the clamped pair (0, 0) by
//     (1, 0): identical for finite logits.
// Requires H == 4 h and W == 4 w (every configuration of the repository); other ratios stay on the per-frame kernels.
#include <hip/hip_runtime.h>

#include "../../include/slotvps_hip.h"

namespace svps {
namespace ppc {

constexpr int ST = SVPS_PPC_STATE_INTS;
constexpr int LDS_PAIR_K = 120;             // K x K pair table kept in LDS by the decide kernel up to this K (57.6 KB)

struct Args {
    const float* masks;        // [T, Ks, h, w] low-resolution logits of the kept slots, descending score order per frame
    long long frame_stride;    // Ks * h * w
    int T, h, w, H, W;
    int* state;                // [T, ST]
    int* pairs;                // [T, pair_stride], frame t: [K, K] (pairs[i * K + j], i < j), zeroed by the caller
    int pair_stride;
    uint8_t* cand;             // [T, H * W, 2]
    uint8_t* out_ids;          // [T, H * W]
    float thr;
    double frac;
    int small_option, stuff_num;
    int rows;                  // frame_stride / (h * w): slot rows a frame of `masks` holds. A frame whose K exceeds it (the caller decoded only the
                               // first `rows` slots of the score order) is SKIPPED by the kernels that read masks - its state is then meaningless
                               // and the caller, who sees K > rows after its one wait, runs the clip again with all rows
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

struct Cell {                  // one thread's 4 x 4 output block: source offsets of its 3 x 3 patch and the eight weight pairs
    int r[3], c[3];            // row offsets (row * w) and columns, clamped
    float x0[4], x1[4], y0[4], y1[4];
};

__device__ __forceinline__ Cell make_cell(int p, int q, int h, int w, int H, int W) {
    Cell ce;
    ce.r[0] = (p > 0 ? p - 1 : 0) * w; ce.r[1] = p * w; ce.r[2] = (p + 1 < h ? p + 1 : h - 1) * w;
    ce.c[0] = q > 0 ? q - 1 : 0; ce.c[1] = q; ce.c[2] = q + 1 < w ? q + 1 : w - 1;
    int i0, i1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        axis_taps(4 * q + i, W, w, i0, i1, ce.x0[i], ce.x1[i]);
        axis_taps(4 * p + i, H, h, i0, i1, ce.y0[i], ce.y1[i]);
    }
    return ce;
}

// the sixteen upsampled logits of one slot: u[4 * s + i] = output (4p + s, 4q + i)
__device__ __forceinline__ void up16(const float* __restrict__ m, const Cell& ce, float* __restrict__ u) {
    float a[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) a[r][c] = m[ce.r[r] + ce.c[c]];
    float hz[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        hz[r][0] = a[r][0] * ce.x0[0] + a[r][1] * ce.x1[0];
        hz[r][1] = a[r][0] * ce.x0[1] + a[r][1] * ce.x1[1];
        hz[r][2] = a[r][1] * ce.x0[2] + a[r][2] * ce.x1[2];
        hz[r][3] = a[r][1] * ce.x0[3] + a[r][2] * ce.x1[3];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int rt = s < 2 ? 0 : 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) u[4 * s + i] = ce.y0[s] * hz[rt][i] + ce.y1[s] * hz[rt + 1][i];
    }
}

// ---- the score filter :684-691 and the descending-score order :580 of every frame. scores / classes [T, L] = softmax(logits).max(-1)
// (computed by the caller); writes K, the kept slots in score order (A_SLOT, and `index` [T, L] int64 for the decode, padded with slot 0),
// their scores, classes and thing flags. Equal scores among kept slots have no defined order in the reference (np.argsort's default
// sort is unstable: its AVX-512 and scalar builds order ties differently); every path of this library takes the order of numpy's
// scalar path, `argsort(kind="stable")[::-1]`: ties in DESCENDING slot order. grid T, 256 threads, L <= 255.
__global__ __launch_bounds__(256) void select_kernel(const float* __restrict__ scores, const long long* __restrict__ classes, int L,
                                                     int drop_last_class, int nc, int num_stuff, float thr, long long* __restrict__ index,
                                                     int* __restrict__ state) {
    __shared__ float s_sc[256];
    __shared__ int s_keep[256], s_k;
    const int tid = threadIdx.x, t = blockIdx.x;
    int* st = state + (size_t)t * ST;
    float sc = 0.f;
    long long cl = 0;
    bool keep = false;
    if (tid < L) {
        sc = scores[(size_t)t * L + tid];
        cl = classes[(size_t)t * L + tid];
        keep = sc > thr && (!drop_last_class || cl != nc - 1);
    }
    s_sc[tid] = sc;
    s_keep[tid] = keep;
    if (tid == 0) s_k = 0;
    __syncthreads();
    int rank = 0;
    if (keep)
        for (int j = 0; j < L; ++j)
            if (s_keep[j] && j != tid) rank += s_sc[j] > sc || (s_sc[j] == sc && j > tid);
    if (keep) atomicAdd(&s_k, 1);
    if (tid < L) index[(size_t)t * L + tid] = 0;
    __syncthreads();
    if (keep) {
        index[(size_t)t * L + rank] = tid;
        st[SVPS_PPC_SLOT + rank] = tid;
        st[SVPS_PPC_SCORE + rank] = __float_as_int(sc);
        st[SVPS_PPC_CL + rank] = (int)cl;
        st[SVPS_PPC_THING + rank] = cl > num_stuff - 1;                 // :594
    }
    if (tid == 0) st[SVPS_PPC_K] = s_k;
}

// ---- candidates of every frame (pp_candidates_kernel per 4 x 4 block). grid (ceil(w / 64), ceil(h / 4), T), 256 threads = 64 x 4 cells
__global__ __launch_bounds__(256) void candidates_kernel(Args a) {
    __shared__ int lcount[256];
    __shared__ uint8_t s_thing[256];
    const int tid = threadIdx.x, t = blockIdx.z;
    int* st = a.state + (size_t)t * ST;
    const int K = st[SVPS_PPC_K];
    if (K > a.rows) return;                      // (uniform per workgroup; see Args::rows)
    lcount[tid] = 0;
    s_thing[tid] = tid < K ? (uint8_t)st[SVPS_PPC_THING + tid] : 0;
    __syncthreads();
    const int q = blockIdx.x * 64 + (tid & 63), p = blockIdx.y * 4 + (tid >> 6);
    const bool on = q < a.w && p < a.h;
    const int hw = a.h * a.w;
    const float* masks = a.masks + (size_t)t * a.frame_stride;
    int key[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) key[i] = -1;
    if (on) {
        const Cell ce = make_cell(p, q, a.h, a.w, a.H, a.W);
        float u[16], mx[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) mx[i] = -INFINITY;
        const float* m = masks;
        for (int k = 0; k < K; ++k, m += hw) {
            up16(m, ce, u);
#pragma unroll
            for (int i = 0; i < 16; ++i) mx[i] = fmaxf(mx[i], u[i]);
        }
        // softmax over the K slots exactly as the reference evaluates it: e_k = exp(u_k - max), p_k = e_k / sum; only the two
        // largest thing terms can reach a threshold > 1/3
        // (selects, no branches: with per-pixel branches hipcc keeps the sixteen-entry arrays as vectors and copies them around)
        float sum[16], e0[16], e1[16];
        int k0[16], k1[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { sum[i] = 0.f; e0[i] = -1.f; e1[i] = -1.f; k0[i] = 255; k1[i] = 255; }
        m = masks;
        for (int k = 0; k < K; ++k, m += hw) {
            up16(m, ce, u);
            const bool thing = s_thing[k] != 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float e = expf(u[i] - mx[i]);
                sum[i] += e;
                const bool g0 = thing && e > e0[i];
                const bool g1 = thing && !g0 && e > e1[i];
                e1[i] = g0 ? e0[i] : (g1 ? e : e1[i]);
                k1[i] = g0 ? k0[i] : (g1 ? k : k1[i]);
                e0[i] = g0 ? e : e0[i];
                k0[i] = g0 ? k : k0[i];
            }
        }
        uint8_t* cand = a.cand + (size_t)t * a.H * a.W * 2;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            unsigned pk[2] = {0u, 0u};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int px = 4 * s + i;
                const bool p0 = k0[px] != 255 && e0[px] / sum[px] >= a.thr, p1 = k1[px] != 255 && e1[px] / sum[px] >= a.thr;
                const int ca = p0 ? k0[px] : 255, cb = p1 ? k1[px] : 255;
                int c0 = ca < cb ? ca : cb;                    // score order = index order
                int c1 = ca < cb ? cb : ca;
                if (c0 == 255) c1 = 255;
                pk[i >> 1] |= (unsigned)(c0 | c1 << 8) << (16 * (i & 1));
                if (c0 != 255) atomicAdd(&lcount[c0], 1);
                if (c1 != 255) { atomicAdd(&lcount[c1], 1); key[px] = c0 << 8 | c1; }
            }
            *reinterpret_cast<uint2*>(cand + 2 * ((size_t)(4 * p + s) * a.W + 4 * q)) = make_uint2(pk[0], pk[1]);
        }
    }
    // pair counts: wave-aggregated - lanes holding the same (c0, c1) pair elect one atomicAdd
    int* pairs = a.pairs + (size_t)t * a.pair_stride;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        unsigned long long todo = __ballot(key[i] >= 0);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int lk = __shfl(key[i], leader);
            const unsigned long long same = __ballot(key[i] == lk);
            if ((tid & 63) == leader) atomicAdd(&pairs[(lk >> 8) * K + (lk & 255)], __popcll(same));
            todo &= ~same;
        }
    }
    __syncthreads();
    if (tid < K && lcount[tid]) atomicAdd(&st[SVPS_PPC_COUNTS + tid], lcount[tid]);
}

// ---- mask_removal :601-640 on the tables (stuff kept first, then things by descending score), keep_inds order, get_ids_area(dedup) :759
__global__ __launch_bounds__(256) void decide_kernel(Args a) {
    __shared__ int s_pairs[LDS_PAIR_K * LDS_PAIR_K];
    __shared__ int s_thing[256], s_cl[256], s_counts[256], s_kept[256], s_cur[256], s_lut[256], s_foc[256];
    __shared__ int s_n, s_ident;
    const int tid = threadIdx.x, t = blockIdx.x;
    int* st = a.state + (size_t)t * ST;
    const int K = st[SVPS_PPC_K];
    const int* gp = a.pairs + (size_t)t * a.pair_stride;
    const bool in_lds = K <= LDS_PAIR_K;
    if (in_lds)
        for (int i = tid; i < K * K; i += 256) s_pairs[i] = gp[i];
    s_thing[tid] = tid < K ? st[SVPS_PPC_THING + tid] : 0;
    s_cl[tid] = tid < K ? st[SVPS_PPC_CL + tid] : 0;
    s_counts[tid] = tid < K ? st[SVPS_PPC_COUNTS + tid] : 0;
    s_kept[tid] = tid < K ? !s_thing[tid] : 0;
    s_foc[tid] = -1;
    __syncthreads();
    if (tid < 64) {                                                   // one wave: sequential over the things, the sum over j across lanes
        const long long n_px = (long long)a.H * a.W;
        for (int i = 0; i < K; ++i) {
            if (!s_thing[i]) continue;
            const int n_i = s_counts[i];
            if (n_i == 0 || n_i == n_px) continue;                    // logit.max() == logit.min() / mask_sum == 0
            long long part = 0;
            for (int j = tid; j < i; j += 64)
                if (s_thing[j] && s_kept[j] && s_cl[j] == s_cl[i]) part += in_lds ? s_pairs[j * K + i] : gp[j * K + i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
            if ((double)part / (double)n_i > a.frac) continue;
            if (tid == 0) s_kept[i] = 1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    if (tid == 0) {
        int n = 0;
        for (int i = 0; i < K; ++i)
            if (!s_thing[i]) s_cur[n++] = i;
        for (int i = 0; i < K; ++i)
            if (s_thing[i] && s_kept[i]) s_cur[n++] = i;
        int ident = 1;
        for (int j = 0; j < n; ++j) {
            const int i = s_cur[j];
            int l = j;
            if (!s_thing[i]) {
                const int c = s_cl[i] & 255;
                if (s_foc[c] < 0) s_foc[c] = j;
                l = s_foc[c];
            }
            s_lut[j] = l;
            ident &= l == j;
        }
        s_n = n;
        s_ident = ident;
    }
    __syncthreads();
    st[SVPS_PPC_KEPT + tid] = s_kept[tid];
    st[SVPS_PPC_CUR + tid] = tid < s_n ? s_cur[tid] : 0;
    st[SVPS_PPC_LUT + tid] = tid < s_n ? s_lut[tid] : 0;
    st[SVPS_PPC_HIST + tid] = 0;
    if (tid == 0) {
        st[SVPS_PPC_N] = s_n;
        st[SVPS_PPC_PHASE] = 0;
        st[SVPS_PPC_ROUNDS] = 0;
        st[SVPS_PPC_LUT_IDENT] = s_ident;
    }
}

// ---- per-pixel first-max argmax over the current slot list of the masks AFTER removal (pp_argmax_kernel per 4 x 4 block).
// IDS = false: histogram of lut[argmax] into the state (frames in phase 0 / 1); IDS = true: ids = lut2[argmax] (frames in phase 2)
template <bool IDS>
__global__ __launch_bounds__(256) void argmax_kernel(Args a) {
    __shared__ int lhist[256];
    __shared__ uint8_t s_sel[256], s_thing[256], s_lut[256], s_kept[256];
    const int tid = threadIdx.x, t = blockIdx.z;
    int* st = a.state + (size_t)t * ST;
    const int phase = st[SVPS_PPC_PHASE];
    if (IDS ? phase != 2 : phase >= 2) return;
    const int n = st[SVPS_PPC_N], K = st[SVPS_PPC_K];
    if (K > a.rows) return;
    lhist[tid] = 0;
    if (tid < n) {
        const int k = st[SVPS_PPC_CUR + tid];
        s_sel[tid] = (uint8_t)k;
        s_thing[tid] = (uint8_t)st[SVPS_PPC_THING + k];
        s_lut[tid] = (uint8_t)st[(IDS ? SVPS_PPC_LUT2 : SVPS_PPC_LUT) + tid];
    }
    s_kept[tid] = tid < K ? (uint8_t)st[SVPS_PPC_KEPT + tid] : 0;
    __syncthreads();
    const int q = blockIdx.x * 64 + (tid & 63), p = blockIdx.y * 4 + (tid >> 6);
    if (q < a.w && p < a.h) {
        const Cell ce = make_cell(p, q, a.h, a.w, a.H, a.W);
        const int hw = a.h * a.w;
        const float* masks = a.masks + (size_t)t * a.frame_stride;
        const uint8_t* cand = a.cand + (size_t)t * a.H * a.W * 2;
        int claimer[16];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const uint2 pk = *reinterpret_cast<const uint2*>(cand + 2 * ((size_t)(4 * p + s) * a.W + 4 * q));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned v = (i < 2 ? pk.x : pk.y) >> (16 * (i & 1));
                const int c0 = v & 255, c1 = (v >> 8) & 255;
                int cl = 255;
                if (c0 != 255 && s_kept[c0]) cl = c0;
                else if (c1 != 255 && s_kept[c1]) cl = c1;
                claimer[4 * s + i] = cl;
            }
        }
        float best[16], u[16];
        int bj[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { best[i] = -INFINITY; bj[i] = 0; }
        for (int j = 0; j < n; ++j) {
            const int k = s_sel[j];
            const bool thing = s_thing[j] != 0;
            bool any = !thing;
            if (thing) {
#pragma unroll
                for (int i = 0; i < 16; ++i) any |= claimer[i] == k;
            }
            if (any) up16(masks + (size_t)k * hw, ce, u);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float v = 0.f;
                if (!thing || k == claimer[i]) v = u[i];
                if (v > best[i]) { best[i] = v; bj[i] = j; }      // first maximum wins (torch.argmax)
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            unsigned pk = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int id = n > 0 ? s_lut[bj[4 * s + i]] : 0;
                pk |= (unsigned)id << (8 * i);
                if (!IDS) atomicAdd(&lhist[id], 1);
            }
            if (IDS) *reinterpret_cast<unsigned*>(a.out_ids + (size_t)t * a.H * a.W + (size_t)(4 * p + s) * a.W + 4 * q) = pk;
        }
    }
    if (!IDS) {
        __syncthreads();
        if (lhist[tid]) atomicAdd(&st[SVPS_PPC_HIST + tid], lhist[tid]);
    }
}

// ---- one step of the small-area loop :760-790 / the relabel table of simple_test :420-433, per frame, on the histogram just taken
__global__ __launch_bounds__(256) void step_kernel(Args a) {
    __shared__ int s_hist[256], s_cur[256], s_thing[256], s_cl[256], s_lut[256], s_area[256], s_lut2[256];
    __shared__ int s_hdr[8];
    const int tid = threadIdx.x, t = blockIdx.x;
    int* st = a.state + (size_t)t * ST;
    const int phase = st[SVPS_PPC_PHASE];
    if (phase >= 2) return;
    const int n0 = st[SVPS_PPC_N];
    s_hist[tid] = st[SVPS_PPC_HIST + tid];
    const int k = tid < n0 ? st[SVPS_PPC_CUR + tid] : 0;
    s_cur[tid] = k;
    s_thing[tid] = tid < n0 ? st[SVPS_PPC_THING + k] : 0;
    s_cl[tid] = tid < n0 ? st[SVPS_PPC_CL + k] : 0;
    s_lut[tid] = 0;
    s_lut2[tid] = 0;
    s_area[tid] = tid < n0 ? st[SVPS_PPC_AREA + tid] : 0;
    __syncthreads();
    if (tid == 0) {
        int n = n0, ph = phase, ident = st[SVPS_PPC_LUT_IDENT], rounds = st[SVPS_PPC_ROUNDS], zero_hist = 0, new_lut = 0;
        bool finalize = false;
        if (ph == 0) {
            for (int j = 0; j < n; ++j) s_area[j] = s_hist[j];
            int m = 0;
            for (int j = 0; j < n; ++j) {
                const int ar = s_area[j];
                bool small;
                if (a.small_option == 0) small = ar <= 4;
                else if (a.small_option == 1) small = s_thing[j] ? ar < 256 : ar < 4;
                else small = !s_thing[j] ? ar < 4096 : ar < 256;
                if (!small) { s_cur[m] = s_cur[j]; s_thing[m] = s_thing[j]; s_cl[m] = s_cl[j]; ++m; }
            }
            if (m < n) {                                   // some segment is small: the survivors go round again with the identity table
                n = m;
                ident = 1;
                new_lut = 1;
                zero_hist = 1;
                ++rounds;
                if (n == 0) {                              // `while len(cur) > 0`: nothing left
                    for (int j = 0; j < 256; ++j) s_hist[j] = 0;
                    finalize = true;
                }
            } else if (ident) {
                finalize = true;                           // the area histogram IS the "which positions own pixels" histogram
            } else {
                ph = 1;                                    // stuff de-duplication was in the table: one more pass with the identity table
                ident = 1;
                new_lut = 1;
                zero_hist = 1;
            }
        } else {
            finalize = true;                               // phase 1: the histogram of the identity pass
        }
        if (finalize) {
            // simple_test :420-433: positions that own pixels, things numbered from the back, stuff by POSITION in unique()
            int inst = 0;
            for (int j = 0; j < n; ++j) inst += s_thing[j] != 0;
            int np = 0;
            for (int j = 0; j < n; ++j)
                if (s_hist[j] > 0) s_lut[np++] = j;        // s_lut reused as the `present` list
            int count = inst;
            for (int pos = np - 1; pos >= 0; --pos) {
                const int oid = s_lut[pos];
                if (oid >= n - inst) { s_lut2[oid] = a.stuff_num + count - 1; --count; }
                else s_lut2[oid] = s_cl[pos];
            }
            ph = 2;
        }
        s_hdr[0] = n; s_hdr[1] = ph; s_hdr[2] = ident; s_hdr[3] = rounds; s_hdr[4] = zero_hist; s_hdr[5] = new_lut; s_hdr[6] = finalize;
    }
    __syncthreads();
    const int n = s_hdr[0];
    st[SVPS_PPC_CUR + tid] = tid < n ? s_cur[tid] : 0;
    st[SVPS_PPC_AREA + tid] = tid < n ? s_area[tid] : 0;
    if (s_hdr[5]) st[SVPS_PPC_LUT + tid] = tid < n ? tid : 0;
    if (s_hdr[4]) st[SVPS_PPC_HIST + tid] = 0;
    if (s_hdr[6]) st[SVPS_PPC_LUT2 + tid] = s_lut2[tid];
    if (tid == 0) {
        st[SVPS_PPC_N] = n;
        st[SVPS_PPC_PHASE] = s_hdr[1];
        st[SVPS_PPC_LUT_IDENT] = s_hdr[2];
        st[SVPS_PPC_ROUNDS] = s_hdr[3];
    }
}

}  // namespace ppc
}  // namespace svps

extern "C" int svps_panoptic_clip_state_ints(void) { return SVPS_PPC_STATE_INTS; }

extern "C" int svps_panoptic_clip_select(const float* scores, const long long* classes, int T, int L, int nc, int num_classes,
                                         int num_stuff, float threshold, long long* index, int* state, void* stream_) {
    if (!scores || !classes || !index || !state) return SVPS_ERR_BAD_ARG;
    if (T <= 0 || L <= 0 || L > 255 || nc <= 0) return SVPS_ERR_BAD_SHAPE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(svps::ppc::select_kernel, dim3(T), dim3(256), 0, stream, scores, classes, L, nc == num_classes - 1 ? 0 : 1, nc,
                       num_stuff, threshold, index, state);
    return (int)hipGetLastError();
}

extern "C" int svps_panoptic_clip(const float* masks, long long frame_stride, int T, int h, int w, int H, int W, int* state,
                                  int* pairs, int pair_stride, uint8_t* cand, uint8_t* out_ids, float pixel_threshold,
                                  double fraction_threshold, int small_option, int stuff_num, int rounds, int stages,
                                  void* stream_) {
    using namespace svps::ppc;
    if (!masks || !state || !pairs || !cand || !out_ids) return SVPS_ERR_BAD_ARG;
    if (T <= 0 || h <= 0 || w <= 0 || H != 4 * h || W != 4 * w || frame_stride < (long long)h * w || pair_stride < 1) return SVPS_ERR_BAD_SHAPE;
    if (!(pixel_threshold > 1.f / 3.f) || small_option < 0 || small_option > 2 || rounds < 0 || stuff_num < 0 || stuff_num > 255)
        return SVPS_ERR_BAD_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    Args a{masks, frame_stride, T, h, w, H, W, state, pairs, pair_stride, cand, out_ids, pixel_threshold, fraction_threshold,
           small_option, stuff_num, (int)(frame_stride / ((long long)h * w))};
    const dim3 grid((w + 63) / 64, (h + 3) / 4, T);
    svps_prof_mark(SVPS_KERNEL_PANOPTIC_POST, 0, stream);
    if (stages & 1) {
        hipLaunchKernelGGL(candidates_kernel, grid, dim3(256), 0, stream, a);
        hipLaunchKernelGGL(decide_kernel, dim3(T), dim3(256), 0, stream, a);
    }
    if (stages & 2)
        for (int r = 0; r < rounds; ++r) {
            hipLaunchKernelGGL(argmax_kernel<false>, grid, dim3(256), 0, stream, a);
            hipLaunchKernelGGL(step_kernel, dim3(T), dim3(256), 0, stream, a);
        }
    if (stages & 4) hipLaunchKernelGGL(argmax_kernel<true>, grid, dim3(256), 0, stream, a);
    svps_prof_mark(SVPS_KERNEL_PANOPTIC_POST, 1, stream);
    return (int)hipGetLastError();
}
