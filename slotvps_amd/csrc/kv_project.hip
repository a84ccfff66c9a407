// K3 - key / value producer of the slot <-> pixel retriever for gfx950.
//
// Replaces, for all T frames of a stage in one launch, the two pixel-side projections of
// MaskDynamicConv.forward (mmdet/models/detectors/dynamic_mask_head.py:428-433):
//
//     k = norm_k(to_k(features + pos))      :432   (with_pos_embed, :575-583)
//     v = norm_v(to_v(features))            :433
//
// and writes them as the bf16 pixel-major [T, HW, 256] tensors K1 streams. Storage policy (what is
// rounded to bf16): the operand features + pos, the two weight matrices, and the LayerNorm outputs.
// Accumulation, bias, LayerNorm statistics and affine are fp32.
//
// The sine embedding is separable (position_encoding.py:251-255: channels [0,128) depend on the row,
// [128,256) on the column), so the kernel takes the two small tables pos_y [H,128], pos_x [W,128]
// instead of the [HW,256] map: the feature map is the only per-pixel input stream.
//
// Mapping: 8 waves, two per SIMD. Wave (proj, ob) = (w >> 2, w & 3) owns output channels
// [64 ob, 64 ob + 64) of projection proj (0 = k, 1 = v): its 64 x 256 weight block lives in 128 VGPRs
// as MFMA A fragments for the whole kernel; pixels stream through in 32-pixel tiles exactly as in K1
// (LDS-DMA ring, swizzled rows, pixel = MFMA column = lane). Per tile and wave: 32 MFMA 32x32x16.
// LayerNorm over the 256 output channels of a pixel = in-lane sums + lane^32 + one (sum, sum of
// squares) exchange between the four waves of a projection. Results are transposed through an LDS
// out-tile so that HBM sees whole 512-byte pixel rows. Waves 0-3 issue all LDS-DMA, waves 4-7 all
// global stores, so each wave's vmcnt queue holds one kind of operation and the DMA ring is waited
// for with an exact count. Two workgroup barriers per tile.
//
// Roofline: HBM - reads 512 B and writes 1024 B per pixel (1.5 KB/px); 262 kFLOP/px on the matrix
// cores (arithmetic intensity 175 flop/B, below the ~300 flop/B ridge of the chip).
#include "common.h"
#include "../../include/slotvps_hip.h"

namespace svps {

constexpr int kProjAhead = 3;            // feature tiles in flight
constexpr int kProjNF = kProjAhead + 1;  // feature ring depth

struct ProjLds {
    static constexpr int fring = 0;                              // kProjNF x 16 KiB
    static constexpr int xk = kProjNF * kTileBytes;              // bf16(f + pos) tile
    static constexpr int outk = xk + kTileBytes;                 // bf16 k rows of the tile
    static constexpr int outv = outk + kTileBytes;
    static constexpr int stats = outv + kTileBytes;              // [2][4][32] float2
    static constexpr int affine = stats + 2 * 4 * 32 * 8;        // bk bv gk bk' gv bv' : 6 x 256 fp32
    static constexpr int total = affine + 6 * kD * 4;
};

__device__ __forceinline__ u32x4 make_srd_p(const void* base, uint32_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    u32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
    d[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
    d[2] = __builtin_amdgcn_readfirstlane(bytes);
    d[3] = 0x00020000u;
    return d;
}

// see slot_attn.hip: asm so that hipcc does not drain the DMA ring before every LDS read
__device__ __forceinline__ void dma16_srd_p(u32x4 srd, uint32_t lds_addr, int voff, int soff) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_addr), "v"(voff), "s"(srd), "s"(soff)
        : "memory");
}

__device__ __forceinline__ float half_swap_add(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <bool HAS_POS>
__global__ __launch_bounds__(512) void kv_project_kernel(
    const __bf16* __restrict__ feat,    // [T, HW, 256]
    const float* __restrict__ pos_y,    // [H, 128] or null
    const float* __restrict__ pos_x,    // [W, 128] or null
    const __bf16* __restrict__ wk,      // [256, 256] bf16, row = output channel (nn.Linear.weight layout)
    const __bf16* __restrict__ wv,
    const float* __restrict__ bk, const float* __restrict__ bv,
    const float* __restrict__ gk, const float* __restrict__ bek,   // norm_k weight / bias
    const float* __restrict__ gv, const float* __restrict__ bev,   // norm_v weight / bias
    float eps_k, float eps_v,
    __bf16* __restrict__ k_out, __bf16* __restrict__ v_out,        // [T, HW, 256]
    int HW, int W, int tiles_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using Lds = ProjLds;
    constexpr int A = kProjAhead;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int proj = w >> 2, ob = w & 3;
    const int r_ = lane & 31, h_ = lane >> 5;
    const int r = r_, h = h_;
    const int t = blockIdx.y, c = blockIdx.x;

    const int px_begin = c * tiles_per_chunk * kTilePx;
    int px_end = px_begin + tiles_per_chunk * kTilePx;
    px_end = px_end < HW ? px_end : HW;
    const int nt = (px_end - px_begin + kTilePx - 1) / kTilePx;

    float* aff = reinterpret_cast<float*>(smem + Lds::affine);
    for (int i = tid; i < kD; i += 512) {
        aff[i] = bk[i];
        aff[kD + i] = bv[i];
        aff[2 * kD + i] = gk[i];
        aff[3 * kD + i] = bek[i];
        aff[4 * kD + i] = gv[i];
        aff[5 * kD + i] = bev[i];
    }

    // ---- this wave's 64 x 256 weight block -> bf16 A fragments in registers -------------------
    bf16x8 wf[2][16];
    {
        const __bf16* wsrc = proj ? wv : wk;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const __bf16* row = wsrc + (size_t)(64 * ob + 32 * b + r) * kD + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
                wf[b][ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(row + 16 * ks));
        }
    }
    wait_vm<0>();

    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>((SVPS_LDS const void*)smem);
    const u32x4 frs = make_srd_p(feat + (size_t)t * HW * kD, (uint32_t)HW * kRowBytes);
    // waves 0-3: four DMA pieces (8 pixel rows) of every feature tile each
    int voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * ob + 2 * i + h;
        voff[i] = row * kRowBytes + (((lane & 31) ^ swz(row)) * 16);
    }
    auto stage_f = [&](int tile) {
        if (tile >= nt) return;
        const uint32_t st = __builtin_amdgcn_readfirstlane(lds0 + Lds::fring + (tile % kProjNF) * kTileBytes + ob * 4096);
        const int px0 = px_begin + tile * kTilePx;
        const int soff = __builtin_amdgcn_readfirstlane(px0 * kRowBytes);
        if (px0 + kTilePx <= HW) {
#pragma unroll
            for (int i = 0; i < 4; ++i) dma16_srd_p(frs, st + i * 1024, voff[i], soff);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 8 * ob + 2 * i + h;
                const int src = px0 + row < HW ? row : HW - 1 - px0;
                dma16_srd_p(frs, st + i * 1024, src * kRowBytes + (((lane & 31) ^ swz(row)) * 16), soff);
            }
        }
    };

    // xk(tile) = bf16(f(tile) + pos), built by waves 4-7 only (thread -> pixel, 32 channels): the DMA
    // waves 0-3 must not execute compiler-visible vector-memory loads inside the loop - hipcc's waits
    // for those would drain the asm DMA ring with them (vmcnt retires in order).
    auto build_xk = [&](int tile) {
        if constexpr (!HAS_POS) return;
        if (w < 4) return;
        int lt = tid - 256;
        asm volatile("" : "+v"(lt));     // opaque: keeps hipcc from hoisting (and then spilling) the addresses
        const int xpx = lt >> 3, xc = lt & 7;                     // channels 32 xc .. 32 xc + 31
        const char* ft = smem + Lds::fring + (tile % kProjNF) * kTileBytes;
        char* xt = smem + Lds::xk;
        int p = px_begin + tile * kTilePx + xpx;
        p = p < HW ? p : HW - 1;
        const int y = p / W, x = p - y * W;
        const float* ptab = xc < 4 ? pos_y + (size_t)y * 128 + 32 * xc : pos_x + (size_t)x * 128 + 32 * (xc - 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int off = xpx * kRowBytes + (((4 * xc + u) ^ swz(xpx)) * 16);
            const bf16x8 f = *reinterpret_cast<const bf16x8*>(ft + off);
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(ptab + 8 * u);
            const f32x4 p1 = *reinterpret_cast<const f32x4*>(ptab + 8 * u + 4);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = (__bf16)((float)f[j] + p0[j]);
                o[4 + j] = (__bf16)((float)f[4 + j] + p1[j]);
            }
            *reinterpret_cast<bf16x8*>(xt + off) = o;
        }
    };

    // out tile -> HBM by waves 4-7: 32 KiB per tile, 8 x 16 B per thread, whole 512-B rows per piece
    auto store_out = [&](int tile) {
        int lt = tid - 256;                             // 0..255
        asm volatile("" : "+v"(lt));
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int piece = u * 256 + lt;             // 0..2047 : [tensor][row][chunk position]
            const int tens = piece >> 10, row = (piece >> 5) & 31, cpos = piece & 31;
            const int px = px_begin + tile * kTilePx + row;
            const u32x4 val = *reinterpret_cast<const u32x4*>(smem + (tens ? Lds::outv : Lds::outk) + row * kRowBytes + cpos * 16);
            if (px < px_end) {
                __bf16* dst = (tens ? v_out : k_out) + ((size_t)t * HW + px) * kD + ((cpos ^ swz(row)) * 8);
                *reinterpret_cast<u32x4*>(dst) = val;
            }
        }
    };

    if (w < 4) {
#pragma unroll
        for (int b = 0; b < A; ++b) stage_f(b);
        if (nt > A - 1) wait_vm<4 * (A - 1)>();
        else wait_vm<0>();
    }
    wg_barrier();          // f(0) landed, affine table written
    build_xk(0);

    float2* stats = reinterpret_cast<float2*>(smem + Lds::stats) + proj * 4 * 32;
    const float* bias = aff + proj * kD;
    const float* gam = aff + (2 + 2 * proj) * kD;
    const float* bet = aff + (3 + 2 * proj) * kD;
    const float eps = proj ? eps_v : eps_k;
    char* outt = smem + (proj ? Lds::outv : Lds::outk);

    for (int it = 0; it < nt; ++it) {
        wg_barrier();                                                  // a(it): xk(it) built, out(it-1) complete
        if (w < 4) stage_f(it + A);
        else if (it >= 1) store_out(it - 1);
        const char* bt = (HAS_POS && proj == 0) ? smem + Lds::xk : smem + Lds::fring + (it % kProjNF) * kTileBytes;
        int r = r_, h = h_;
        asm volatile("" : "+v"(r), "+v"(h));   // opaque per iteration: no loop-invariant address tables in VGPRs

        f32x16 acc[2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {     // 4 feature fragments in flight (register budget: 128 of 256 hold W)
            bf16x8 xf[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) xf[u] = read_row_frag(bt, 4 * grp + u, r, h);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0][4 * grp + u], xf[u], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1][4 * grp + u], xf[u], acc[1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // + bias; LayerNorm statistics of this wave's 64 channels for pixel column r
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {            // channels 64 ob + 32 b + 8 g + 4 h + (0..3)
                const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + 64 * ob + 32 * b + 8 * g + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x = acc[b][4 * g + j] + bb[j];
                    acc[b][4 * g + j] = x;
                    s1 += x;
                    s2 = fmaf(x, x, s2);
                }
                __builtin_amdgcn_sched_barrier(0);   // keep hipcc from pre-loading every table row (VGPR budget)
            }
        s1 = half_swap_add(s1);
        s2 = half_swap_add(s2);
        if (h == 0) stats[ob * 32 + r] = make_float2(s1, s2);
        if (w < 4) {                                                   // f(it+1) landed for the DMA waves
            if (it + A < nt) wait_vm<4 * (A - 1)>();
            else wait_vm<0>();
        }
        wg_barrier();                                                  // b(it)
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
            const float2 sv = stats[ww * 32 + r];
            t1 += sv.x;
            t2 += sv.y;
        }
        const float mean = t1 * (1.f / kD);
        const float var = fmaxf(t2 * (1.f / kD) - mean * mean, 0.f);
        const float rstd = rsqrtf(var + eps);
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 o;
                const int ch0 = 64 * ob + 32 * b + 8 * g + 4 * h;     // acc_row(4g + j, h) = 8g + 4h + j
                const f32x4 gg = *reinterpret_cast<const f32x4*>(gam + ch0);
                const f32x4 be = *reinterpret_cast<const f32x4*>(bet + ch0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    o[j] = (__bf16)((acc[b][4 * g + j] - mean) * rstd * gg[j] + be[j]);
                const int chunk = ch0 >> 3;                            // 16-byte chunk of the pixel row
                *reinterpret_cast<bf16x4*>(outt + r * kRowBytes + ((chunk ^ swz(r)) * 16) + (ch0 & 7) * 2) = o;
                __builtin_amdgcn_sched_barrier(0);
            }
        if (it + 1 < nt) build_xk(it + 1);
    }
    wg_barrier();
    if (w >= 4) store_out(nt - 1);
}

}  // namespace svps

namespace {
int proj_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return n;
}
}  // namespace

extern "C" int svps_kv_project_fwd(const void* feat, const float* pos_y, const float* pos_x, const void* wk,
                                   const float* bk, const float* lnk_w, const float* lnk_b, float lnk_eps,
                                   const void* wv, const float* bv, const float* lnv_w, const float* lnv_b,
                                   float lnv_eps, void* k_out, void* v_out, int T, int H, int W, int D,
                                   void* stream_) {
    if (!feat || !wk || !bk || !lnk_w || !lnk_b || !wv || !bv || !lnv_w || !lnv_b || !k_out || !v_out)
        return SVPS_ERR_BAD_ARG;
    if ((pos_y == nullptr) != (pos_x == nullptr)) return SVPS_ERR_BAD_ARG;
    if (D != svps::kD || T <= 0 || H <= 0 || W <= 0) return SVPS_ERR_BAD_SHAPE;
    const int HW = H * W;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int tiles = (HW + svps::kTilePx - 1) / svps::kTilePx;
    int chunks = proj_num_cus() / T;
    if (chunks < 1) chunks = 1;
    if (chunks > tiles) chunks = tiles;
    const int tpc = (tiles + chunks - 1) / chunks;
    chunks = (tiles + tpc - 1) / tpc;
    static bool attr_set[2] = {false, false};
    const bool has_pos = pos_y != nullptr;
    auto kern = has_pos ? svps::kv_project_kernel<true> : svps::kv_project_kernel<false>;
    if (!attr_set[has_pos]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, svps::ProjLds::total);
        if (e != hipSuccess) return (int)e;
        attr_set[has_pos] = true;
    }
    svps_prof_mark(SVPS_KERNEL_KV_PROJECT, 0, stream);
    hipLaunchKernelGGL(kern, dim3(chunks, T), dim3(512), svps::ProjLds::total, stream,
                       static_cast<const __bf16*>(feat), pos_y, pos_x, static_cast<const __bf16*>(wk),
                       static_cast<const __bf16*>(wv), bk, bv, lnk_w, lnk_b, lnv_w, lnv_b,
                       lnk_eps, lnv_eps, static_cast<__bf16*>(k_out), static_cast<__bf16*>(v_out), HW, W, tpc);
    svps_prof_mark(SVPS_KERNEL_KV_PROJECT, 1, stream);
    return (int)hipGetLastError();
}
