// K3-HL - both LayerNorm statistics of the fused retriever at the reference's precision from ONE read of the hi / lo planes (round 4).
//
// MaskDynamicConv.forward (mmdet/models/detectors/dynamic_mask_head.py:432-433) normalises k = to_k(f + pos) and v = to_v(f) per pixel;
// the fused retriever (retr_attn.hip) only needs the two reciprocal standard deviations: var = |R x + r|^2 / 256 with [W~ | b~] = Q [R | r]
// (retr_stats.hip). In the reference-precision mode (precision "fp16x2") the map is f = hi + lo (two fp16 planes, level_fuse_hl.hip) and
// both factors are R = hi + lo: R x = R_hi x_hi + R_lo x_hi + R_hi x_lo, three MFMAs per k-step into one fp32 accumulator.
//
// The first form of this path (retr_stats_t.hip's HL template: one launch per projection, register-staged tiles one ahead) read the
// planes twice and had ONE tile in flight per CU: 41.6 ms of a 160-frame step, 2.7 TB/s - the latency of a tile's loads, not a resource.
// Here: four waves of 512 registers (one per SIMD). Wave j holds row blocks j and 7 - j (R is upper triangular: 18 k-steps together) of
// ALL FOUR factor matrices (key / value x hi / lo: 288 registers) and runs the key chain, then the value chain, on the same staged tile
// (54 + 54 MFMAs per tile); tiles are fetched TWO ahead through registers (plain loads, compiler-counted waits) into a double-buffered
// padded LDS tile pair. The position term of the key statistics and the constant columns r arrive as fp32 tables in ACCUMULATOR order
// (the host permutes the 256 columns so that a lane's 16 values of a row block are 64 contiguous bytes): Ty' [H, 256] = Ty + r_k,
// Tx' [W, 256], r_v' [256]. One barrier per tile; wave 0 finishes the previous tile from the four waves' sums and writes the whole
// 16-byte aux rows of retr_stats.hip: {1, hi sigma_v, lo sigma_v, 0 (fp16), rstd_k, rstd_v (fp32)}.
#include "common.h"
#include "../../include/slotvps_hip.h"

#ifndef SVPS_SHL_ABL
#define SVPS_SHL_ABL 0      // timing-only ablations (wrong results): 1 no table loads, 2 no MFMAs, 4 no global tile loads, 8 no finish
#endif

namespace svps {

typedef __attribute__((ext_vector_type(8))) _Float16 sh_f16x8;

constexpr int kShRow = 256 * 2 + 16;                 // staged pixel row: 256 fp16 + pad (conflict-free 16-byte fragment reads)
struct StatsHlLds {
    static constexpr int plane = kTilePx * kShRow;   // one plane of one tile
    static constexpr int xt = 0;                     // [2 buffers][hi, lo][32 px][528 B]
    static constexpr int part = 4 * plane;           // [2 buffers][key, value][4 waves][32 px] float
    static constexpr int total = part + 2 * 2 * 4 * 32 * 4;
};

struct StatsHlArgs {
    const _Float16* f_hi;      // [T, HW, 256]
    const _Float16* f_lo;
    const _Float16* rk_hi;     // [256, 256] upper triangular factors, hi and lo parts
    const _Float16* rk_lo;
    const _Float16* rv_hi;
    const _Float16* rv_lo;
    const float* tyk;          // [ty_rows, 256]  Ty + r_k in accumulator order (ty_rows = H, or 1: no position term)
    const float* txk;          // [tx_rows, 256]  Tx in accumulator order (tx_rows = W, or 1: a zero row)
    const float* rbv;          // [256]  r_v in accumulator order
    _Float16* aux;             // [T, HW, 8]
    float eps_k, eps_v;
    int HW, W, ty_rows, tx_rows, tiles_per_wg;
};

// accumulator order: column 32 RB + 16 h + 4 g + j of a table row <-> factor row 32 RB + 8 g + 4 h + j (register 4 g + j of lane half h)
template <int J>
__device__ __forceinline__ void stats_hl_role(const StatsHlArgs& a, char* smem, int lane) {
    constexpr int KA = 2 * J, NA = 16 - KA;          // row block J: k-steps KA .. 15
    constexpr int KB = 14 - 2 * J, NB = 16 - KB;     // row block 7 - J: k-steps KB .. 15
    constexpr int RBA = J, RBB = 7 - J;
    const int r = lane & 31, h = lane >> 5;
    const int t = blockIdx.y;
    const int tid = threadIdx.x;                     // 0 .. 255
    sh_f16x8 kha[NA], kla[NA], khb[NB], klb[NB], vha[NA], vla[NA], vhb[NB], vlb[NB];
    {
        const size_t ra = (size_t)(32 * RBA + r) * 256 + 8 * h, rb = (size_t)(32 * RBB + r) * 256 + 8 * h;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            kha[i] = *reinterpret_cast<const sh_f16x8*>(a.rk_hi + ra + 16 * (KA + i));
            kla[i] = *reinterpret_cast<const sh_f16x8*>(a.rk_lo + ra + 16 * (KA + i));
            vha[i] = *reinterpret_cast<const sh_f16x8*>(a.rv_hi + ra + 16 * (KA + i));
            vla[i] = *reinterpret_cast<const sh_f16x8*>(a.rv_lo + ra + 16 * (KA + i));
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            khb[i] = *reinterpret_cast<const sh_f16x8*>(a.rk_hi + rb + 16 * (KB + i));
            klb[i] = *reinterpret_cast<const sh_f16x8*>(a.rk_lo + rb + 16 * (KB + i));
            vhb[i] = *reinterpret_cast<const sh_f16x8*>(a.rv_hi + rb + 16 * (KB + i));
            vlb[i] = *reinterpret_cast<const sh_f16x8*>(a.rv_lo + rb + 16 * (KB + i));
        }
    }
    const int tiles = (a.HW + kTilePx - 1) / kTilePx;
    const int tile0 = blockIdx.x * a.tiles_per_wg;
    int tile1 = tile0 + a.tiles_per_wg;
    tile1 = tile1 < tiles ? tile1 : tiles;
    const _Float16* FH = a.f_hi + (size_t)t * a.HW * 256;
    const _Float16* FL = a.f_lo + (size_t)t * a.HW * 256;
    float* part = reinterpret_cast<float*>(smem + StatsHlLds::part);
    // staging: thread -> (pixel tid >> 3, 64 bytes = 32 channels) of both planes
    const int spx = tid >> 3, sc = tid & 7;
    struct Pre { u32x4 h[4], l[4]; };
    auto fetch = [&](int tile, Pre& p) {
        int gp = tile * kTilePx + spx;
        gp = gp < a.HW ? gp : a.HW - 1;              // ragged last tile / past the chunk: a valid pixel (not stored)
        const size_t o = (size_t)gp * 256 + 32 * sc;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            p.h[i] = *reinterpret_cast<const u32x4*>(FH + o + 8 * i);
            p.l[i] = *reinterpret_cast<const u32x4*>(FL + o + 8 * i);
        }
    };
    auto stage = [&](int buf, const Pre& p) {
        char* dst = smem + StatsHlLds::xt + buf * 2 * StatsHlLds::plane + spx * kShRow + 64 * sc;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32x4*>(dst + 16 * i) = p.h[i];
            *reinterpret_cast<u32x4*>(dst + StatsHlLds::plane + 16 * i) = p.l[i];
        }
    };
    // wave 0, lanes h == 0: the tile's 32 aux rows from the four waves' sums (one 16-byte store per pixel: whole rows)
    auto finish = [&](int tile) {
        const int cur = (tile - tile0) & 1;
        float tk = 0.f, tv = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
            tk += part[(cur * 2 + 0) * 128 + ww * 32 + r];
            tv += part[(cur * 2 + 1) * 128 + ww * 32 + r];
        }
        const float rk = __builtin_amdgcn_rsqf(tk * (1.f / 256.f) + a.eps_k);
        const float varv = tv * (1.f / 256.f) + a.eps_v;
        const float rv = __builtin_amdgcn_rsqf(varv);
        float sigma = varv * rv;
        asm volatile("" : "+v"(sigma));              // one fp32 value for both halves (see retr_attn.hip, p2_store)
        const _Float16 sh = (_Float16)sigma, sl = (_Float16)(sigma - (float)sh), one = (_Float16)1.0f;
        u32x4 row;
        row[0] = (uint32_t)__builtin_bit_cast(uint16_t, one) | ((uint32_t)__builtin_bit_cast(uint16_t, sh) << 16);
        row[1] = (uint32_t)__builtin_bit_cast(uint16_t, sl);
        row[2] = __float_as_uint(rk);
        row[3] = __float_as_uint(rv);
        const int gp = tile * kTilePx + r;
        if (h == 0 && gp < a.HW) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(a.aux) + ((size_t)t * a.HW + gp) * 16) = row;
    };
    // one chain: 32 factor rows x 32 pixels, k-steps K0 .. 15, accumulator started from `c` (the constant column / position terms)
    auto sumsq = [&](const f32x16& acc) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i] * acc[i];
        return s;
    };

    Pre p0, p1;
    fetch(tile0, p0);
    stage(0, p0);
    if (tile0 + 1 < tile1) fetch(tile0 + 1, p1);
    __syncthreads();
    auto body = [&](int tile, Pre& pn, Pre& pf) {     // pn: holds tile + 1 (staged at the end); pf: receives tile + 2
        const int cur = (tile - tile0) & 1;
        const int px0 = tile * kTilePx;
        // The key chains start from Ty' + Tx' of the lane's pixel (L2-resident tables, 16 loads): requested here, consumed behind the value
        // chain. Timing-only ablations (SVPS_SHL_ABL, finest level, T = 40: 2 790 us): without these loads 2 165, without the MFMAs 1 557,
        // without the tile loads 2 274, without both 1 177 - the parts ADD UP (one in-order wave per SIMD: nothing overlaps), and moving the
        // tables' use behind 54 MFMAs changed nothing (2 820): what they cost is the ISSUE of sixteen 64-sector loads, not their latency.
        // The next step for this kernel is a column-strip tile order (Tx' constant per strip: eight of the sixteen loads disappear).
        int gp = px0 + r;
        gp = gp < a.HW ? gp : a.HW - 1;
        const int y = gp / a.W, x = gp - y * a.W;
        const float* tyr = a.tyk + (size_t)(y < a.ty_rows ? y : a.ty_rows - 1) * 256 + 16 * h;
        const float* txr = a.txk + (size_t)(x < a.tx_rows ? x : a.tx_rows - 1) * 256 + 16 * h;
        f32x4 ya[4], xa[4], yb[4], xb[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#if SVPS_SHL_ABL & 1
            ya[g] = xa[g] = yb[g] = xb[g] = f32x4{0.f, 0.f, 0.f, 0.f};
            (void)tyr; (void)txr;
#else
            ya[g] = *reinterpret_cast<const f32x4*>(tyr + 32 * RBA + 4 * g);
            xa[g] = *reinterpret_cast<const f32x4*>(txr + 32 * RBA + 4 * g);
            yb[g] = *reinterpret_cast<const f32x4*>(tyr + 32 * RBB + 4 * g);
            xb[g] = *reinterpret_cast<const f32x4*>(txr + 32 * RBB + 4 * g);
#endif
        }
        const char* xh = smem + StatsHlLds::xt + cur * 2 * StatsHlLds::plane + r * kShRow + 16 * h;
        const char* xl = xh + StatsHlLds::plane;
        constexpr int K0 = KA < KB ? KA : KB;
        // the loads of tile + 2, two at a time between the k-steps of the value chain (a 1-KiB vector-memory instruction holds its wave
        // for ~100 cycles at issue; spread or in a row measures the same)
        const bool more2 = tile + 2 < tile1 && !(SVPS_SHL_ABL & 4);
        int gp2 = (tile + 2) * kTilePx + spx;
        gp2 = gp2 < a.HW ? gp2 : a.HW - 1;
        const size_t o2 = (size_t)gp2 * 256 + 32 * sc;
        // ---- value side first. Fragments one k-step ahead of their MFMAs; the fences keep hipcc from hoisting all sixteen k-steps' reads
        //      (128 registers the weights leave no room for) to the top of the tile
        f32x16 va, vb;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 ra_ = *reinterpret_cast<const f32x4*>(a.rbv + 32 * RBA + 16 * h + 4 * g);
            const f32x4 rb_ = *reinterpret_cast<const f32x4*>(a.rbv + 32 * RBB + 16 * h + 4 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                va[4 * g + j] = ra_[j];
                vb[4 * g + j] = rb_[j];
            }
        }
        sh_f16x8 fh[2], fl[2];
        fh[K0 & 1] = *reinterpret_cast<const sh_f16x8*>(xh + 32 * K0);
        fl[K0 & 1] = *reinterpret_cast<const sh_f16x8*>(xl + 32 * K0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = K0; ks < 16; ++ks) {
            if (ks + 1 < 16) {
                fh[(ks + 1) & 1] = *reinterpret_cast<const sh_f16x8*>(xh + 32 * (ks + 1));
                fl[(ks + 1) & 1] = *reinterpret_cast<const sh_f16x8*>(xl + 32 * (ks + 1));
            }
            if (ks >= KA && !(SVPS_SHL_ABL & 2)) {
                va = __builtin_amdgcn_mfma_f32_32x32x16_f16(vla[ks - KA], fh[ks & 1], va, 0, 0, 0);
                va = __builtin_amdgcn_mfma_f32_32x32x16_f16(vha[ks - KA], fl[ks & 1], va, 0, 0, 0);
                va = __builtin_amdgcn_mfma_f32_32x32x16_f16(vha[ks - KA], fh[ks & 1], va, 0, 0, 0);
            }
            if (ks >= KB && !(SVPS_SHL_ABL & 2)) {
                vb = __builtin_amdgcn_mfma_f32_32x32x16_f16(vlb[ks - KB], fh[ks & 1], vb, 0, 0, 0);
                vb = __builtin_amdgcn_mfma_f32_32x32x16_f16(vhb[ks - KB], fl[ks & 1], vb, 0, 0, 0);
                vb = __builtin_amdgcn_mfma_f32_32x32x16_f16(vhb[ks - KB], fh[ks & 1], vb, 0, 0, 0);
            }
            const int fi = (ks - K0 - 1) / 2;                         // loads of tile + 2: piece fi behind k-steps K0 + 1, + 3, + 5, + 7
            if (more2 && ks > K0 && ((ks - K0 - 1) & 1) == 0 && fi < 4) {
                pf.h[fi] = *reinterpret_cast<const u32x4*>(FH + o2 + 8 * fi);
                pf.l[fi] = *reinterpret_cast<const u32x4*>(FL + o2 + 8 * fi);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const float sv0 = sumsq(va) + sumsq(vb);
        // ---- key side (the same fragments, read again: the registers hold the weights)
        f32x16 ka, kb;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ka[4 * g + j] = ya[g][j] + xa[g][j];
                kb[4 * g + j] = yb[g][j] + xb[g][j];
            }
        fh[K0 & 1] = *reinterpret_cast<const sh_f16x8*>(xh + 32 * K0);
        fl[K0 & 1] = *reinterpret_cast<const sh_f16x8*>(xl + 32 * K0);
#pragma unroll
        for (int ks = K0; ks < 16; ++ks) {
            if (ks + 1 < 16) {
                fh[(ks + 1) & 1] = *reinterpret_cast<const sh_f16x8*>(xh + 32 * (ks + 1));
                fl[(ks + 1) & 1] = *reinterpret_cast<const sh_f16x8*>(xl + 32 * (ks + 1));
            }
            if (ks >= KA && !(SVPS_SHL_ABL & 2)) {
                ka = __builtin_amdgcn_mfma_f32_32x32x16_f16(kla[ks - KA], fh[ks & 1], ka, 0, 0, 0);
                ka = __builtin_amdgcn_mfma_f32_32x32x16_f16(kha[ks - KA], fl[ks & 1], ka, 0, 0, 0);
                ka = __builtin_amdgcn_mfma_f32_32x32x16_f16(kha[ks - KA], fh[ks & 1], ka, 0, 0, 0);
            }
            if (ks >= KB && !(SVPS_SHL_ABL & 2)) {
                kb = __builtin_amdgcn_mfma_f32_32x32x16_f16(klb[ks - KB], fh[ks & 1], kb, 0, 0, 0);
                kb = __builtin_amdgcn_mfma_f32_32x32x16_f16(khb[ks - KB], fl[ks & 1], kb, 0, 0, 0);
                kb = __builtin_amdgcn_mfma_f32_32x32x16_f16(khb[ks - KB], fh[ks & 1], kb, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const float sk0 = sumsq(ka) + sumsq(kb);
        float sk = sk0, sv = sv0;
        sk += __shfl_xor(sk, 32);
        sv += __shfl_xor(sv, 32);
        if (h == 0) {
            part[(cur * 2 + 0) * 128 + J * 32 + r] = sk;
            part[(cur * 2 + 1) * 128 + J * 32 + r] = sv;
        }
        if (tile + 1 < tile1) stage(cur ^ 1, pn);
        __syncthreads();                             // sums of this tile complete; the next tile is staged
        if (J == 0 && !(SVPS_SHL_ABL & 8)) finish(tile);
    };
    for (int tile = tile0; tile < tile1; tile += 2) {
        body(tile, p1, p0);
        if (tile + 1 < tile1) body(tile + 1, p0, p1);
    }
}

__global__ __launch_bounds__(256) void retr_stats_hl_kernel(StatsHlArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {
        case 0: stats_hl_role<0>(a, smem, lane); break;
        case 1: stats_hl_role<1>(a, smem, lane); break;
        case 2: stats_hl_role<2>(a, smem, lane); break;
        default: stats_hl_role<3>(a, smem, lane); break;
    }
}

}  // namespace svps

// svps_retr_stats_hl_fwd (include/slotvps_hip.h): tyk [ty_rows, 256] = Ty + r_k, txk [tx_rows, 256] = Tx, rbv [256] - fp32, columns in
// ACCUMULATOR order (column 32 B + 16 h + 4 g + j = factor row 32 B + 8 g + 4 h + j); ty_rows = H or 1, tx_rows = W or 1 (no position term:
// one row each, txk zero).
extern "C" int svps_retr_stats_hl_fwd(const void* feat_hi, const void* feat_lo, const float* tyk, int ty_rows, const float* txk, int tx_rows,
                                      const void* rk_hi, const void* rk_lo, float lnk_eps, const void* rv_hi, const void* rv_lo,
                                      const float* rbv, float lnv_eps, void* aux, int T, int H, int W, int D, void* stream_) {
    if (!feat_hi || !feat_lo || !tyk || !txk || !rk_hi || !rk_lo || !rv_hi || !rv_lo || !rbv || !aux) return SVPS_ERR_BAD_ARG;
    if (D != svps::kD || T <= 0 || H <= 0 || W <= 0) return SVPS_ERR_BAD_SHAPE;
    if ((ty_rows != H && ty_rows != 1) || (tx_rows != W && tx_rows != 1)) return SVPS_ERR_BAD_SHAPE;
    if ((size_t)H * W > svps::kMaxFramePixels) return SVPS_ERR_BAD_SHAPE;
    const int HW = H * W;
    const int tiles = (HW + svps::kTilePx - 1) / svps::kTilePx;
    int chunks = svps_pick_chunks(T, tiles, svps_num_cus());
    const int tpw = (tiles + chunks - 1) / chunks;
    chunks = (tiles + tpw - 1) / tpw;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    using H16 = _Float16;
    static SvpsLdsAttr attr;
    if (hipError_t ae = attr.ensure(reinterpret_cast<const void*>(svps::retr_stats_hl_kernel), svps::StatsHlLds::total); ae != hipSuccess) return (int)ae;
    const svps::StatsHlArgs args{static_cast<const H16*>(feat_hi), static_cast<const H16*>(feat_lo), static_cast<const H16*>(rk_hi),
                                 static_cast<const H16*>(rk_lo), static_cast<const H16*>(rv_hi), static_cast<const H16*>(rv_lo), tyk, txk, rbv,
                                 static_cast<H16*>(aux), lnk_eps, lnv_eps, HW, W, ty_rows, tx_rows, tpw};
    svps_prof_mark(SVPS_KERNEL_RETR_STATS, 0, stream);
    hipLaunchKernelGGL(svps::retr_stats_hl_kernel, dim3(chunks, T), dim3(256), svps::StatsHlLds::total, stream, args);
    svps_prof_mark(SVPS_KERNEL_RETR_STATS, 1, stream);
    return (int)hipGetLastError();
}
