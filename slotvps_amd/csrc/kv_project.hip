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
// out-tile so that HBM sees whole 512-byte pixel rows. The key waves issue the feature LDS-DMA, the value waves the position rows; every wave stores
// its own projection's rows; the DMA ring is waited for with exact vmcnt counts. Two workgroup barriers per tile.
//
// The position rows of a tile (ytab[y(p)], xtab[x(p)], fp32) also arrive by LDS-DMA, gathered with
// per-lane source addresses, so that building the key operand bf16(f + pos) is pure LDS + VALU work:
// a compiler-visible global load in the loop would make hipcc wait on vmcnt, which retires in order
// and therefore also waits for every global store of the previous tile.
//
// Roofline: HBM - reads 512 B and writes 1024 B per pixel (1.5 KB/px); 262 kFLOP/px on the matrix
// cores (arithmetic intensity 175 flop/B, below the ~300 flop/B ridge of the chip).
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "../../include/slotvps_hip.h"

namespace svps {

constexpr int kProjAhead = 2;            // feature tiles in flight (the vector L1 returns in order and an HBM miss takes
                                         // about 0.7 tile times: two tiles ahead are enough, and the LDS of the fourth
                                         // ring slot buys the second position buffer)
constexpr int kProjNF = kProjAhead + 1;  // feature ring depth

struct ProjLds {
    static constexpr int fring = 0;                              // kProjNF x 16 KiB
    static constexpr int xk = kProjNF * kTileBytes;              // bf16(f + pos) tile
    // out tiles: rows padded to 528 B instead of swizzled - every LDS address of the LayerNorm epilogue is then
    // "lane base + compile-time constant" (no per-write address arithmetic), at a two-way bank conflict on the 8-byte writes
    static constexpr int kOutRow = kRowBytes + 16;
    static constexpr int kOutTile = kTilePx * kOutRow;           // 16.5 KiB
    static constexpr int outk = xk + kOutTile;                   // bf16 k rows of the tile (xk is padded the same way)
    static constexpr int outv = outk + kOutTile;
    // position rows. W % 32 == 0 (a tile lies inside one image row): two buffers of 32 xtab rows (16 KiB each) and of one
    // ytab row (512 B each), requested TWO tiles ahead - the vector L1 returns loads in order, so these L2 hits come
    // back behind the feature tiles (HBM misses) requested before them and a request for the next tile would land late.
    // Other widths: one buffer of per-pixel ytab rows (16 KiB) and one of xtab rows, requested one tile ahead.
    static constexpr int posx = outv + kOutTile;                 // [2][32 px][128] fp32 | ytab rows, then xtab rows
    static constexpr int posy = posx + 2 * kTileBytes;           // [2][256] fp32 (aligned tiles only; one 64-lane DMA = 1 KiB, the row is its first half)
    static constexpr int posx_of(bool aligned, int tile) { return posx + (aligned ? (tile & 1) * kTileBytes : kTileBytes); }
    static constexpr int posy_of(bool aligned, int tile) { return aligned ? posy + (tile & 1) * 1024 : posx; }
    static constexpr int stats = posy + 2048;              // [2][4][32] float2
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
    uint32_t keep;      // M0 (LDS base of the DMA) is saved and restored inside the statement
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

// Four 1-KiB pieces with one M0 write: the instruction offset advances BOTH the global address and the LDS
// address (LDS = M0 + inst_offset + lane * 16), so piece i lands at lds_addr + 1024 i and reads from
// voff_i + soff + 1024 i - callers pre-subtract 1024 i from voff_i where the source is not contiguous.
__device__ __forceinline__ void dma16x4_srd_p(u32x4 srd, uint32_t lds_addr, int v0, int v1, int v2, int v3, int soff) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %2, %6, %7 offen lds\n\t"
        "buffer_load_dwordx4 %3, %6, %7 offen offset:1024 lds\n\t"
        "buffer_load_dwordx4 %4, %6, %7 offen offset:2048 lds\n\t"
        "buffer_load_dwordx4 %5, %6, %7 offen offset:3072 lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_addr), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(srd), "s"(soff)
        : "memory");
}

__device__ __forceinline__ float half_swap_add(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

#ifdef SVPS_K3_STAMP
__device__ unsigned long long k3_stamps[8][8][16];      // [wave][iteration - 8][point]
#define K3_STAMP(pt)                                                                              \
    do {                                                                                          \
        if (blockIdx.x == 7 && blockIdx.y == 0 && it >= 8 && it < 16 && lane == 0)               \
            k3_stamps[w][it - 8][pt] = __builtin_amdgcn_s_memtime();                              \
    } while (0)
#else
#define K3_STAMP(pt) do {} while (0)
#endif

template <bool HAS_POS, int ABL = 0>
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
        if constexpr (ABL & 4) return;
        const uint32_t st = __builtin_amdgcn_readfirstlane(lds0 + Lds::fring + (tile % kProjNF) * kTileBytes + ob * 4096);
        const int px0 = px_begin + tile * kTilePx;
        const int soff = __builtin_amdgcn_readfirstlane(px0 * kRowBytes);
        if (px0 + kTilePx <= HW) {
            dma16x4_srd_p(frs, st, voff[0], voff[1] - 1024, voff[2] - 2048, voff[3] - 3072, soff);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 8 * ob + 2 * i + h;
                const int src = px0 + row < HW ? row : HW - 1 - px0;
                dma16_srd_p(frs, st + i * 1024, src * kRowBytes + (((lane & 31) ^ swz(row)) * 16), soff);
            }
        }
    };

    // position rows of tile `tile` -> LDS, by waves 0-3: wave ob covers pixel rows 8 ob .. 8 ob + 7; one DMA
    // instruction moves two 512-byte table rows (lanes 0-31: pixel 2i, lanes 32-63: pixel 2i + 1).
    const u32x4 ysrd = make_srd_p(pos_y, HAS_POS ? (uint32_t)((HW + W - 1) / W) * 512u : 0u);
    const u32x4 xsrd = make_srd_p(pos_x, HAS_POS ? (uint32_t)W * 512u : 0u);
    const bool aligned_rows = (W & 31) == 0;
    auto stage_pos = [&](int tile) {
        if constexpr (!HAS_POS) return;
        if (tile >= nt) return;
        if constexpr (ABL & 4) return;
        const uint32_t sy = __builtin_amdgcn_readfirstlane(lds0 + Lds::posy_of(aligned_rows, tile) + ob * 4096);
        const uint32_t sx = __builtin_amdgcn_readfirstlane(lds0 + Lds::posx_of(aligned_rows, tile) + ob * 4096);
        if (aligned_rows) {
            // W % 32 == 0: a tile lies inside one image row. One 512-byte ytab row for the whole tile (build_xk reads
            // it for every pixel) and 32 consecutive xtab rows = one contiguous 16 KiB block.
            const int px0 = px_begin + tile * kTilePx;
            const int y0 = __builtin_amdgcn_readfirstlane(px0 / W), x0 = px0 - y0 * W;
            if (ob == 0) dma16_srd_p(ysrd, __builtin_amdgcn_readfirstlane(lds0 + Lds::posy_of(true, tile)), lane * 16, y0 * 512);
            const int v = ob * 4096 + lane * 16;
            dma16x4_srd_p(xsrd, sx, v, v, v, v, __builtin_amdgcn_readfirstlane(x0 * 512));
            return;
        }
        const int p = px_begin + tile * kTilePx + 8 * ob + h;
        const int yy = p / W, xx = p - yy * W;                 // rows past the image read zeros (buffer bounds)
        int yo = yy * 512 + (lane & 31) * 16, xo = xx * 512 + (lane & 31) * 16;
        const int xwrap = W * 512 + (lane & 31) * 16;          // xo of column W
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dma16_srd_p(ysrd, sy + i * 1024, yo, 0);
            dma16_srd_p(xsrd, sx + i * 1024, xo, 0);
            xo += 1024;                                        // two pixels on; at most two row wraps (W == 1)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const bool wrap = xo >= xwrap;
                xo = wrap ? xo - W * 512 : xo;
                yo = wrap ? yo + 512 : yo;
            }
        }
    };

    // xk(tile) = bf16(f(tile) + pos(tile)), by waves 0-3 from LDS only: thread -> 16-byte chunk (8 channels).
    auto build_xk = [&](int tile) {
        if constexpr (!HAS_POS) return;
        if (w >= 4) return;
        int lt = tid;
        asm volatile("" : "+v"(lt));     // opaque: keeps hipcc from hoisting (and then spilling) the addresses
        // chunk u of this thread = 16-byte chunk cpos of pixel row 8u + q. Everything is "one lane base + u * constant":
        // swz(8u + q) = swz(q) ^ (2 if u is odd), i.e. the swizzled source chunk of odd u is the even one with address
        // bit 5 flipped; the position row advances by 8 rows per u (not at all for the single ytab row of an aligned tile).
        const int q = lt >> 5, cpos = lt & 31;
        const int fe = Lds::fring + (tile % kProjNF) * kTileBytes + q * kRowBytes + ((cpos ^ swz(q)) << 4);
        const int fo = fe ^ 32;
        const bool ypart = cpos < 16;
        const int pb = (ypart ? Lds::posy_of(aligned_rows, tile) + (aligned_rows ? 0 : q * 512)
                              : Lds::posx_of(aligned_rows, tile) + q * 512) + (cpos & 15) * 32;
        const int ps = (ypart && aligned_rows) ? 0 : 8 * 512;
        const int xo = Lds::xk + q * Lds::kOutRow + cpos * 16;                  // xk rows are padded, not swizzled
        bf16x8 fv[4];
        f32x4 pv[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {                          // twelve LDS reads in flight, then the arithmetic
            fv[u] = *reinterpret_cast<const bf16x8*>(smem + ((u & 1) ? fo : fe) + u * 8 * kRowBytes);
            const char* pt = smem + pb + u * ps;
            pv[u][0] = *reinterpret_cast<const f32x4*>(pt);
            pv[u][1] = *reinterpret_cast<const f32x4*>(pt + 16);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = (__bf16)((float)fv[u][j] + pv[u][0][j]);
                o[4 + j] = (__bf16)((float)fv[u][4 + j] + pv[u][1][j]);
            }
            *reinterpret_cast<bf16x8*>(smem + xo + u * 8 * Lds::kOutRow) = o;
        }
    };

    // out tile of this wave's projection -> HBM: 16 KiB per tile, 4 x 16 B per thread. The global side is linear
    // (thread -> consecutive 16-byte pieces of the tile's 16 KiB, scalar tile offset, hardware bounds check at the end
    // of the frame), the LDS side undoes the row swizzle.
    const __amdgpu_buffer_rsrc_t osrd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((proj ? v_out : k_out) + (size_t)t * HW * kD), 0, (ABL & 1) ? 0 : px_end * kRowBytes, 0x00020000);
    auto store_out = [&](int tile) {
        int lt = tid & 255;
        asm volatile("" : "+v"(lt));
        const char* src = smem + (proj ? Lds::outv : Lds::outk);
        // The tile offset goes into the VGPR offset, not into soffset: the hardware range check covers voffset only,
        // and with an SGPR soffset hipcc omits the wait state gfx950 needs between a >64-bit buffer store and the next
        // VALU write of its data registers (observed: first data dword of the first store clobbered).
        const int base = (px_begin + tile * kTilePx) * kRowBytes + lt * 16;
        u32x4 val[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = 8 * u + (lt >> 5), gc = lt & 31;           // global chunk gc of pixel row `row`
            val[u] = *reinterpret_cast<const u32x4*>(src + row * Lds::kOutRow + gc * 16);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            __builtin_amdgcn_raw_buffer_store_b128(val[u], osrd, base + u * 4096, 0, 0);
    };

    // DMA issue is split between the roles: the value waves gather the position rows, the key waves (which have slack
    // at the end of their matrix half) stream the feature tiles.
    const int pa = aligned_rows ? 2 : 1;                                 // position rows: tiles ahead
    const int npos = aligned_rows ? (ob == 0 ? 5 : 4) : 8;              // DMA instructions of one stage_pos() of this wave
    if (w >= 4) {
        stage_pos(0);
        if (pa == 2) stage_pos(1);
        wait_vm<0>();
    } else {
#pragma unroll
        for (int b = 0; b < A; ++b) stage_f(b);
        wait_vm_dyn(4 * (1 < nt));                   // f(0) is the oldest
    }
    wg_barrier();          // f(0), pos(0) landed, affine table written
    build_xk(0);

    float2* stats = reinterpret_cast<float2*>(smem + Lds::stats) + proj * 4 * 32;
    const float* bias = aff + proj * kD;
    const float* gam = aff + (2 + 2 * proj) * kD;
    const float* bet = aff + (3 + 2 * proj) * kD;
    const float eps = proj ? eps_v : eps_k;
    f32x16 acc[2];
    // bias of this lane's 32 accumulator rows, resident (acc_row(4g + j, h) = 8g + 4h + j): the accumulators start from it
    f32x16 breg[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + 64 * ob + 32 * b + 8 * g + 4 * h_);
#pragma unroll
            for (int j = 0; j < 4; ++j) breg[b][4 * g + j] = bb[j];
        }

    // "heavy" half of a tile: store the previous out tile, 32 MFMA, + bias, LayerNorm partial sums -> LDS
    auto heavy = [&](int it, auto padded_tag) {
        constexpr bool PADDED = decltype(padded_tag)::value;      // B operand rows padded to 528 B (xk) or swizzled (feature ring)
        const char* bt = (HAS_POS && proj == 0) ? smem + Lds::xk : smem + Lds::fring + (it % kProjNF) * kTileBytes;
        int r = r_, h = h_;
        asm volatile("" : "+v"(r), "+v"(h));   // opaque per iteration: no loop-invariant address tables in VGPRs
        // accumulators start from the bias of their channels (acc_row(4g + j, h) = 8g + 4h + j): eight 16-byte LDS
        // reads straight into the accumulator registers, in flight together with the first operand fragments
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[b][i] = (ABL & 64) ? 0.f : breg[b][i];
        // swizzled rows: the XOR with swz(r) < 16 only touches the low four bits of the chunk index 2ks + h, so k-steps
        // ks and ks + 8 differ by a constant 256 B: eight computed offsets serve all sixteen fragments
        int o8[8];
        if constexpr (!PADDED) {
            const int s4 = swz(r);
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) o8[k8] = r * kRowBytes + (((2 * k8 + h) ^ s4) << 4);
        }
        auto frag = [&](int ks) {
            if constexpr (PADDED) return *reinterpret_cast<const bf16x8*>(bt + r * Lds::kOutRow + (2 * ks + h) * 16);
            else return *reinterpret_cast<const bf16x8*>(bt + o8[ks & 7] + (ks >> 3) * 256);
        };
        if constexpr (!(ABL & 2)) {
            bf16x8 xf[2][4];                     // operand fragments, double-buffered in groups of four k-steps
#pragma unroll
            for (int u = 0; u < 4; ++u) xf[0][u] = frag(u);
#pragma unroll
            for (int grp = 0; grp < 4; ++grp) {
                if (grp < 3) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) xf[(grp + 1) & 1][u] = frag(4 * (grp + 1) + u);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0][4 * grp + u], xf[grp & 1][u], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1][4 * grp + u], xf[grp & 1][u], acc[1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        K3_STAMP(4);
        f32x2 p1 = {0.f, 0.f}, p2 = {0.f, 0.f};       // packed fp32 (v_pk_add_f32 / v_pk_fma_f32): 2 elements per VALU slot
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const f32x2 x = {acc[b][i], acc[b][i + 1]};
                p1 += x;
                p2 = __builtin_elementwise_fma(x, x, p2);
            }
        float s1 = p1[0] + p1[1], s2 = p2[0] + p2[1];
        s1 = half_swap_add(s1);
        s2 = half_swap_add(s2);
        if (h == 0) stats[ob * 32 + r] = make_float2(s1, s2);
        K3_STAMP(5);
    };

    // "light" half: LayerNorm of the accumulators with the exchanged statistics -> bf16 out tile
    auto light = [&](int it) {
        int r = r_, h = h_;
        asm volatile("" : "+v"(r), "+v"(h));
        // out-tile write address = lane base + compile-time constant: channel 64 ob + 32 b + 8 g + 4 h of pixel row r
        int wo = (proj ? Lds::outv : Lds::outk) + r * Lds::kOutRow + 8 * h + 128 * ob;
        asm volatile("" : "+v"(wo));
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
        const f32x2 a2 = {rstd, rstd}, b2 = {-mean * rstd, -mean * rstd};     // (x - mean) * rstd = x * a + b
#pragma unroll
        for (int b = 0; b < ((ABL & 8) ? 0 : 2); ++b) {
            f32x4 gg[4], be[4];                                        // eight LDS reads in flight per block
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch0 = 64 * ob + 32 * b + 8 * g + 4 * h;     // acc_row(4g + j, h) = 8g + 4h + j
                if constexpr (ABL & 32) {                              // ablation: affine-free LayerNorm
                    gg[g] = f32x4{1.f, 1.f, 1.f, 1.f};
                    be[g] = f32x4{0.f, 0.f, 0.f, 0.f};
                    continue;
                }
                gg[g] = *reinterpret_cast<const f32x4*>(gam + ch0);
                be[g] = *reinterpret_cast<const f32x4*>(bet + ch0);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const f32x2 x = {acc[b][4 * g + j], acc[b][4 * g + j + 1]};
                    const f32x2 gj = {gg[g][j], gg[g][j + 1]}, bj = {be[g][j], be[g][j + 1]};
                    const f32x2 y = __builtin_elementwise_fma(__builtin_elementwise_fma(x, a2, b2), gj, bj);
                    o[j] = (__bf16)y[0];
                    o[j + 1] = (__bf16)y[1];
                }
                *reinterpret_cast<bf16x4*>(smem + wo + 64 * b + 16 * g) = o;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // Ping-pong: the key waves (0-3) and the value waves (4-7) share every SIMD pairwise and run half a tile
    // apart, so that the matrix pipe of a SIMD works for one of them while the other does its VALU / LDS / store
    // part. Two workgroup barriers per tile:
    //   phase X(it): key waves heavy(it)                       | value waves: all DMA issue, light(it-1)
    //   phase Y(it): key waves light(it), build xk(it+1)       | value waves heavy(it)
    // The two roles run separate loops (same number of barriers): with one loop body for both, hipcc kept two copies of
    // the accumulators and moved 32 registers back and forth per tile. The last half-phases (stores of the last tile,
    // LayerNorm of the last value tile) are peeled so that every iteration redefines the accumulators unconditionally.
    if (nt <= 0) return;
    if (proj == 0) {
        for (int it = 0; it < nt; ++it) {
            K3_STAMP(0);
            wg_barrier();                                              // X(it): xk(it) built, stats_v(it-1) written
            K3_STAMP(1);
            if (it >= 1) store_out(it - 1);
            K3_STAMP(3);
            heavy(it, std::bool_constant<HAS_POS>{});
            K3_STAMP(6);
            stage_f(it + A);
            // f(it+1) (this wave's pieces, issued in X(it+1-A)) has to be in LDS before build_xk(it+1): vmcnt retires in
            // order, younger are the stores of X(it) and f(it+2)
            static_assert(A == 2, "wait count below assumes two tiles in flight");
            wait_vm_dyn(4 * ((it >= 1) + (it + 2 < nt)));
            K3_STAMP(7);
            wg_barrier();                                              // Y(it): stats_k(it) written
            K3_STAMP(8);
            light(it);
            K3_STAMP(9);
            if (!(ABL & 16) && it + 1 < nt) build_xk(it + 1);
            K3_STAMP(10);
        }
        wg_barrier();                                                  // X(nt)
        store_out(nt - 1);
        wg_barrier();                                                  // Y(nt)
    } else {
        for (int it = 0; it < nt; ++it) {
            K3_STAMP(0);
            wg_barrier();                                              // X(it)
            K3_STAMP(1);
            stage_pos(it + pa);                                        // its buffer held pos(it), consumed in Y(it-1)
            K3_STAMP(2);
            if (it >= 1) light(it - 1);
            K3_STAMP(11);
            // pos(it+1) landed. Two tiles ahead: it was requested in X(it-1); younger are the stores of Y(it-1) and pos(it+2)
            if (pa == 2) wait_vm_dyn(4 * (it >= 2) + ((it + 2 < nt) ? npos : 0));
            else wait_vm<0>();
            K3_STAMP(7);
            wg_barrier();                                              // Y(it): out_v(it-1) complete
            K3_STAMP(8);
            if (it >= 1) store_out(it - 1);
            K3_STAMP(3);
            heavy(it, std::false_type{});
            K3_STAMP(10);
        }
        wg_barrier();                                                  // X(nt)
        light(nt - 1);
        wg_barrier();                                                  // Y(nt)
        store_out(nt - 1);
    }
}

}  // namespace svps

namespace {
int proj_num_cus() { return svps_num_cus(); }
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
    if ((size_t)H * W > svps::kMaxFramePixels) return SVPS_ERR_BAD_SHAPE;   // 32-bit buffer offsets inside a frame
    const int HW = H * W;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int tiles = (HW + svps::kTilePx - 1) / svps::kTilePx;
    int chunks = svps_pick_chunks(T, tiles, proj_num_cus());
    const int tpc = (tiles + chunks - 1) / chunks;
    chunks = (tiles + tpc - 1) / tpc;
    static SvpsLdsAttr attr[2];
    const bool has_pos = pos_y != nullptr;
    auto kern = has_pos ? svps::kv_project_kernel<true> : svps::kv_project_kernel<false>;
#ifdef SVPS_K3_ABLATE
    static int abl = -1;
    if (abl < 0) { const char* e = getenv("SVPS_K3_ABLATE"); abl = e ? atoi(e) : 0; }
    if (has_pos) {
        switch (abl) {
            case 1: kern = svps::kv_project_kernel<true, 1>; break;
            case 2: kern = svps::kv_project_kernel<true, 2>; break;
            case 3: kern = svps::kv_project_kernel<true, 3>; break;
            case 4: kern = svps::kv_project_kernel<true, 4>; break;
            case 8: kern = svps::kv_project_kernel<true, 8>; break;
            case 16: kern = svps::kv_project_kernel<true, 16>; break;
            case 9: kern = svps::kv_project_kernel<true, 9>; break;
            case 25: kern = svps::kv_project_kernel<true, 25>; break;
            case 27: kern = svps::kv_project_kernel<true, 27>; break;
            case 31: kern = svps::kv_project_kernel<true, 31>; break;
            case 30: kern = svps::kv_project_kernel<true, 30>; break;
            case 32: kern = svps::kv_project_kernel<true, 32>; break;
            case 64: kern = svps::kv_project_kernel<true, 64>; break;
            case 96: kern = svps::kv_project_kernel<true, 96>; break;
            default: break;
        }
    }
    static bool abl_attr = false;
    if (!abl_attr) { hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, svps::ProjLds::total); abl_attr = true; }
#endif
    if (hipError_t ae = attr[has_pos].ensure(reinterpret_cast<const void*>(kern), svps::ProjLds::total); ae != hipSuccess)
        return (int)ae;
    svps_prof_mark(SVPS_KERNEL_KV_PROJECT, 0, stream);
    hipLaunchKernelGGL(kern, dim3(chunks, T), dim3(512), svps::ProjLds::total, stream,
                       static_cast<const __bf16*>(feat), pos_y, pos_x, static_cast<const __bf16*>(wk),
                       static_cast<const __bf16*>(wv), bk, bv, lnk_w, lnk_b, lnv_w, lnv_b,
                       lnk_eps, lnv_eps, static_cast<__bf16*>(k_out), static_cast<__bf16*>(v_out), HW, W, tpc);
    svps_prof_mark(SVPS_KERNEL_KV_PROJECT, 1, stream);
    return (int)hipGetLastError();
}

#ifdef SVPS_K3_STAMP
extern "C" int svps_k3_debug_read(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(svps::k3_stamps), sizeof(unsigned long long) * 8 * 8 * 16);
}
#endif
