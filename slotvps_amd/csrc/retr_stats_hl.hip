// K3-HL - both LayerNorm statistics of the fused retriever at the reference's precision from ONE read of the hi / lo planes (round 4).
//
// MaskDynamicConv.forward (mmdet/models/detectors/dynamic_mask_head.py:432-433) normalises k = to_k(f + pos) and v = to_v(f) per pixel;
// the fused retriever (retr_attn.hip) only needs the two reciprocal standard deviations: var = |R x + r|^2 / 256 with [W~ | b~] = Q [R | r]
// (retr_stats.hip). In the reference-precision mode (precision "fp16x2") the map is f = hi + lo (two fp16 planes, level_fuse_hl.hip) and
// both factors are R = hi + lo: R x = R_hi x_hi + R_lo x_hi + R_hi x_lo, three MFMAs per k-step into one fp32 accumulator.
//
// The first form of this path (one launch per projection, register-staged tiles one ahead) read the
// planes twice and had ONE tile in flight per CU: 41.6 ms of a 160-frame step, 2.7 TB/s - the latency of a tile's loads, not a resource.
// The second (four waves of 512 registers, all four factor matrices per wave, both chains in turn): 31 ms - its timing-only ablations
// (SVPS_SHL_ABL) showed the parts of a lone in-order wave ADDING UP: matrix chain 1 160 us + position tables 620 + tile loads 515 + rest 420
// of 2 790 at the finest level (T = 40). This form (22.7 ms): EIGHT waves, wave (projection, quarter j) holds row blocks j and 7 - j (R is
// upper triangular: 18 k-steps together) of ITS projection's factor as hi and lo A fragments (144 registers) and runs one chain of 54
// MFMAs per tile on the staged hi / lo planes. The key and the value wave of a quarter share a SIMD and work in PING-PONG (two barriers
// per tile, the schedule of retr_stats.hip): while one runs its chain the other does its light work - start values, DMA requests, the
// finish - so the matrix pipe is never shared and (s_memtime stamps, `make stampshl`, tools/shl_stamps.py) both half-periods are
// ~2 400 cycles of chain (54 MFMAs with LDS operands: 36 cycles each + the sums of squares) beside 1 600 - 2 500 of light work.
//   * tiles arrive by LDS-DMA (asm, two tiles ahead, ring of three, the chunk swizzle of common.h): the key waves bring the hi plane, the
//     value waves the lo plane; no staging registers, no LDS writes. A wave's requests are older than the start-value loads of its next
//     light phase, whose wait (vmcnt retires in order) covers them half a period before the tile is read.
//   * the position term of the key statistics and the constant columns r arrive as fp32 tables in ACCUMULATOR order (the host permutes
//     the 256 columns so that a lane's 16 values of a row block are contiguous; for W % 32 == 0 also TILED, so that a wave's load is one
//     contiguous KiB): Ty' [H, 256] = Ty + r_k, Tx' [W, 256], r_v' [256]. All sixteen loads of a key wave are ONE round trip (the Ty'
//     values land in the accumulators themselves; two fenced groups were 2 x 1 500 cycles of L2 latency on the critical path).
//   * wave (value, 0) finishes the previous tile from the eight waves' sums and writes the whole 16-byte aux rows of retr_stats.hip:
//     {1, hi sigma_v, lo sigma_v, 0 (fp16), rstd_k, rstd_v (fp32)}.
// Round 5 - the TILED-TABLES form (template parameter TLDS; W % 32 == 0 with position tables: every level of the product configurations),
// from the stamps and timing-only ablations of that schedule (22.1 -> 18.4 ms per step):
//   * the start values come through LDS as well: a key wave requests its own 8 KiB of Tx' (once per 32-pixel COLUMN: a workgroup walks its
//     tiles column-major, sequence number s -> row s % H, column s / H) and the 16 chunks of the row's Ty' by LDS-DMA one period ahead and
//     reads back what it requested (no barrier, vmcnt alone); r_v' sits in LDS. No wave waits on an L2 round trip, no staging registers.
//   * each wave's sums of squares moved into ITS light half (the accumulators are free until the next start values arrive from LDS), so
//     a half-period is the bare chain; the finish of a tile follows two half-periods later.
//   * tile + 2 is requested at the TOP of the light half, BEFORE the landing wait of tile + 1 (s_waitcnt vmcnt(4): the four new requests
//     stay in flight): two tiles in flight per CU - with one, the loads took longer than a period to land.
//   * s_setprio 1 around a chain: the value waves are the younger half of the workgroup and lose issue arbitration to the key waves'
//     light work (their chain took 2 150 cycles beside the key waves' 1 910; static priority for the younger half was zero-sum).
// The other form (!TLDS: no position tables, or W % 32 != 0) keeps the schedule described above.
#include "common.h"
#include "../../include/slotvps_hip.h"

#ifndef SVPS_SHL_EXP
#define SVPS_SHL_EXP 0      // precision experiments (separate library only). Bit 0, measured in round 5: the VALUE statistic without its R_lo x_hi term -
                            // 54 -> 36 MFMAs per value wave and tile, K3-HL 22.8 -> 20.3 ms per step (the step 76.2 -> 73.1 ms), but the full-size
                            // parity goes from the reference's own reproducibility (mask logits 8.7e-6, stages <= 2.7e-5) to 2.9e-5 / 4.9e-5 (sharp
                            // case 3.3e-4 -> 8.1e-4): inside the tolerance, not adopted - the mode's point is to be indistinguishable from the reference
#endif
#ifndef SVPS_SHL_AHEAD
#define SVPS_SHL_AHEAD 2    // operand fragments requested this many k-steps ahead of their MFMAs
#endif
#ifndef SVPS_SHL_STAGGER
#define SVPS_SHL_STAGGER 0  // tiled-tables form: 1 = ONE barrier per tile - the key waves run chain then light work, the value waves light work then chain, so
                            // the two chains of a SIMD overlap in the middle of the period (a lone chain issues an MFMA every 36 cycles, two share the pipe at 32)
#endif
#ifndef SVPS_SHL_PRIO
#define SVPS_SHL_PRIO 1     // tiled-tables form: 1 = s_setprio 1 around each chain, 2 = static s_setprio 1 for the value waves (the younger half), 0 none
#endif
#ifndef SVPS_SHL_ABL
#define SVPS_SHL_ABL 0      // timing-only ablations (wrong results): 1 no table loads, 2 no MFMAs, 4 no global tile loads, 8 no finish,
                            // (tiled tables) 16 no start values in the loop, 32 no sums of squares, 64 no table requests in the loop
#endif

namespace svps {

#ifdef SVPS_SHL_STAMP
// diagnostic build only (tools/shl_stamps.py): s_memtime stamps of the key and the value wave of quarter 0 of one workgroup, tiles 8 .. 15
__device__ unsigned long long shl_stamps[2][8][8];           // [key / value][tile - 8][point]
#define SHL_STAMP(pt)                                                                                         \
    do {                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if (blockIdx.x == 3 && blockIdx.y == 2 && J == 0 && tile - tile0 >= 8 && tile - tile0 < 16 && lane == 0) \
            shl_stamps[KEY ? 0 : 1][tile - tile0 - 8][pt] = __builtin_amdgcn_s_memtime();                     \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    } while (0)
#else
#define SHL_STAMP(pt) do {} while (0)
#endif

typedef __attribute__((ext_vector_type(8))) _Float16 sh_f16x8;

struct StatsHlLds {
    static constexpr int kSlots = 3;                 // tile ring: tile t in work, t + 1 landed, t + 2 in flight
    static constexpr int plane = kTileBytes;         // one plane of one tile: 32 pixel rows of 512 B, 16-byte chunks swizzled (common.h)
    static constexpr int xt = 0;                     // [3 slots][hi, lo][16 KiB]
    static constexpr int part = kSlots * 2 * plane;  // [3 buffers][key, value][4 waves][32 px] float (three: the staggered schedule has ONE barrier per tile)
    static constexpr int total = part + 3 * 2 * 4 * 32 * 4;
    // TILED tables only (round 5): the start values come through LDS as well, so that no wave waits on an L2 round trip
    static constexpr int tabx = total;               // [4 key waves][2 row blocks][4 g][1 KiB: lane's 16 bytes]  Tx' of the NEXT tile
    static constexpr int taby = tabx + 4 * 8192;     // [4 key waves][1 KiB: 16 chunks (block, h, g), repeated]     Ty' of the next tile's row
    static constexpr int rbv = taby + 4 * 1024;      // [256] r_v' (constant)
    static constexpr int total_tables = rbv + 1024;
};

// asm LDS-DMA (invisible to hipcc's wait counting: the builtin form is drained with s_waitcnt vmcnt(0) before the next LDS read)
__device__ __forceinline__ u32x4 shl_make_srd(const void* base, uint32_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    u32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
    d[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
    d[2] = __builtin_amdgcn_readfirstlane(bytes);
    d[3] = 0x00020000u;
    return d;
}
__device__ __forceinline__ void shl_dma16(u32x4 srd, uint32_t lds_addr, int voff, int soff) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %2, %3, %4 offen nt lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_addr), "v"(voff), "s"(srd), "s"(soff)
        : "memory");
}

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
    int tx_tiled;              // txk is in the TILED order [W / 32][8 row blocks][4 g][2 h][32 pixels][4] (W % 32 == 0): a wave's load of
                               // (row block, g) for the 32 pixels of a tile is one contiguous KiB instead of 64 separate sectors
};

// accumulator order: column 32 RB + 16 h + 4 g + j of a table row <-> factor row 32 RB + 8 g + 4 h + j (register 4 g + j of lane half h)
// One wave = (projection KEY / value, quarter J): row blocks J and 7 - J of that projection's factor, hi and lo (144 registers).
template <int J, bool KEY, bool TLDS>
__device__ __forceinline__ void stats_hl_role(const StatsHlArgs& a, char* smem, int lane) {
    constexpr int KA = 2 * J, NA = 16 - KA;          // row block J: k-steps KA .. 15
    constexpr int KB = 14 - 2 * J, NB = 16 - KB;     // row block 7 - J: k-steps KB .. 15
    constexpr int RBA = J, RBB = 7 - J;
    constexpr int K0 = KA < KB ? KA : KB;
    const int r = lane & 31, h = lane >> 5;
    const int t = blockIdx.y;
    sh_f16x8 wha[NA], wla[NA], whb[NB], wlb[NB];
    {
        const _Float16* rh = KEY ? a.rk_hi : a.rv_hi;
        const _Float16* rl = KEY ? a.rk_lo : a.rv_lo;
        const size_t ra = (size_t)(32 * RBA + r) * 256 + 8 * h, rb = (size_t)(32 * RBB + r) * 256 + 8 * h;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            wha[i] = *reinterpret_cast<const sh_f16x8*>(rh + ra + 16 * (KA + i));
            wla[i] = *reinterpret_cast<const sh_f16x8*>(rl + ra + 16 * (KA + i));
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            whb[i] = *reinterpret_cast<const sh_f16x8*>(rh + rb + 16 * (KB + i));
            wlb[i] = *reinterpret_cast<const sh_f16x8*>(rl + rb + 16 * (KB + i));
        }
    }
    const int tiles = (a.HW + kTilePx - 1) / kTilePx;
    // (TLDS) a workgroup walks its tiles COLUMN-major - sequence number s -> image row s % H, 32-pixel column s / H - so that the Tx'
    // values of its tiles change once per H tiles instead of every tile (a tile is 16 contiguous KiB either way)
    const int rows = a.HW / a.W, wt = a.W >> 5;
    auto where = [&](int s) { return TLDS ? (s % rows) * wt + s / rows : s; };
    const int tile0 = blockIdx.x * a.tiles_per_wg;
    int tile1 = tile0 + a.tiles_per_wg;
    tile1 = tile1 < tiles ? tile1 : tiles;
    float* part = reinterpret_cast<float*>(smem + StatsHlLds::part);
    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>((SVPS_LDS const void*)smem);
    // staging by LDS-DMA: the key waves bring the hi plane of a tile, the value waves the lo plane - wave quarter J rows 8 J .. 8 J + 7
    // (four 1-KiB pieces of two pixel rows); no registers, no LDS writes, requested two tiles ahead in a ring of three
    const u32x4 srd = shl_make_srd((KEY ? a.f_hi : a.f_lo) + (size_t)t * a.HW * 256, (uint32_t)a.HW * (uint32_t)kRowBytes);
    int voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * J + 2 * i + h;
        voff[i] = row * kRowBytes + ((r ^ swz(row)) * 16);      // (rows past the end of the frame: out of range, read as zeros)
    }
    auto request = [&](int tile) {
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + StatsHlLds::xt + ((tile - tile0) % 3) * 2 * StatsHlLds::plane +
                                                            (KEY ? 0 : StatsHlLds::plane) + J * 4096);
        const int soff = __builtin_amdgcn_readfirstlane(where(tile) * kTileBytes);
#pragma unroll
        for (int i = 0; i < 4; ++i) shl_dma16(srd, dst + i * 1024, voff[i], soff);
    };
    // wave (key, 0), lanes h == 0: the tile's 32 aux rows from the eight waves' sums (one 16-byte store per pixel: whole rows)
    auto finish = [&](int tile) {
        const int cur = (tile - tile0) % 3;
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
        const int gp = where(tile) * kTilePx + r;
        if (h == 0 && gp < a.HW) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(a.aux) + ((size_t)t * a.HW + gp) * 16) = row;
    };
    auto sumsq = [&](const f32x16& acc) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i] * acc[i];
        return s;
    };

    // start values of the two chains of `tile`: the key waves' Ty' + Tx' rows of the lane's pixel (L2-resident tables: 16 loads whose ISSUE
    // costs ~1 000 cycles), the value waves' constant column r_v
    f32x16 ca, cb;
    auto start_values = [&](int tile) {
        if constexpr (KEY) {
            int gp = tile * kTilePx + r;
            gp = gp < a.HW ? gp : a.HW - 1;
            const int y = gp / a.W, x = gp - y * a.W;
            const float* tyr = a.tyk + (size_t)(y < a.ty_rows ? y : a.ty_rows - 1) * 256 + 16 * h;
            // plain order: row x, columns 32 RB + 16 h + 4 g (+ j); tiled order: [x / 32][RB][g][h][x % 32][j] -> the same expression
            // `txr + 32 * RB + 4 * g` below with strides (sb, sg) = (32, 4) resp. (1024, 256) floats
            const int sb = a.tx_tiled ? 1024 : 32, sg = a.tx_tiled ? 256 : 4;
            const float* txr = a.tx_tiled ? a.txk + (size_t)(x >> 5) * 8192 + (h * 32 + (x & 31)) * 4
                                          : a.txk + (size_t)(x < a.tx_rows ? x : a.tx_rows - 1) * 256 + 16 * h;
            // all sixteen loads in ONE round trip (two fenced groups measured 2 x 1 500 cycles of L2 latency on the critical path of the
            // second half-period): the Ty' values land in the accumulators themselves, only the Tx' values need registers of their own
            f32x4 xa[4], xb[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#if SVPS_SHL_ABL & 1
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const f32x4 ya = z, yb = z;
                xa[g] = z; xb[g] = z;
                (void)tyr; (void)txr; (void)sb; (void)sg;
#else
                const f32x4 ya = *reinterpret_cast<const f32x4*>(tyr + 32 * RBA + 4 * g), yb = *reinterpret_cast<const f32x4*>(tyr + 32 * RBB + 4 * g);
                xa[g] = *reinterpret_cast<const f32x4*>(txr + sb * RBA + sg * g);
                xb[g] = *reinterpret_cast<const f32x4*>(txr + sb * RBB + sg * g);
#endif
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ca[4 * g + j] = ya[j];
                    cb[4 * g + j] = yb[j];
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ca[4 * g + j] += xa[g][j];
                    cb[4 * g + j] += xb[g][j];
                }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 ra_ = *reinterpret_cast<const f32x4*>(a.rbv + 32 * RBA + 16 * h + 4 * g);
                const f32x4 rb_ = *reinterpret_cast<const f32x4*>(a.rbv + 32 * RBB + 16 * h + 4 * g);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ca[4 * g + j] = ra_[j];
                    cb[4 * g + j] = rb_[j];
                }
            }
        }
    };
    // the chain of `tile` on the staged planes, then this wave's 64 rows' sum of squares per pixel
    auto heavy = [&](int tile) {
        const int cur = (tile - tile0) % 3;
        // this lane's 16-byte chunk of k-step ks in pixel row r: chunk (2 ks + h) ^ swz(r) - one XOR on the lane's base address
        const uint32_t xh = lds0 + StatsHlLds::xt + ((tile - tile0) % 3) * 2 * StatsHlLds::plane + r * kRowBytes + ((h ^ swz(r)) << 4);
        const uint32_t xl = xh + StatsHlLds::plane;
        auto frag = [](uint32_t base, int ks) {
            return *reinterpret_cast<SVPS_LDS const sh_f16x8*>((uintptr_t)(base ^ ((uint32_t)ks << 5)));
        };
        // fragments TWO k-steps ahead of their MFMAs (a k-step of one row block is three MFMAs = 96 cycles, less than an LDS round trip under
        // load); the fences keep hipcc from hoisting all sixteen k-steps' reads to the top of the tile
        constexpr int AH = SVPS_SHL_AHEAD, NB = AH + 1;         // k-steps ahead (2; 3 and 4 measured in round 6: profiles/r06/README.md)
        sh_f16x8 fh[NB], fl[NB];
#pragma unroll
        for (int k = K0; k < K0 + AH && k < 16; ++k) {
            fh[k % NB] = frag(xh, k);
            fl[k % NB] = frag(xl, k);
        }
        if (TLDS && SVPS_SHL_PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = K0; ks < 16; ++ks) {
            if (ks + AH < 16) {
                fh[(ks + AH) % NB] = frag(xh, ks + AH);
                fl[(ks + AH) % NB] = frag(xl, ks + AH);
            }
            constexpr bool kVLo = KEY || !(SVPS_SHL_EXP & 1);      // EXPERIMENT bit 0: the value statistic without its R_lo x_hi term
            // bits 1 / 2 (round 6): the value / key statistic without its R_hi x_lo term - the MAP's low part, an error that is independent
            // from pixel to pixel. Measured WORSE than bit 0 (mask logits T5 9.5e-6 -> 4.1e-5, VIPER T10 7.7e-5; both: 1.0e-4): at the coarse
            // levels a slot owns ~20 pixels, nothing averages out (profiles/r06/README.md)
            constexpr bool kXLo = KEY ? !(SVPS_SHL_EXP & 4) : !(SVPS_SHL_EXP & 2);
            if (ks >= KA && !(SVPS_SHL_ABL & 2)) {
                if (kVLo) ca = __builtin_amdgcn_mfma_f32_32x32x16_f16(wla[ks - KA], fh[ks % NB], ca, 0, 0, 0);
                if (kXLo) ca = __builtin_amdgcn_mfma_f32_32x32x16_f16(wha[ks - KA], fl[ks % NB], ca, 0, 0, 0);
                ca = __builtin_amdgcn_mfma_f32_32x32x16_f16(wha[ks - KA], fh[ks % NB], ca, 0, 0, 0);
            }
            if (ks >= KB && !(SVPS_SHL_ABL & 2)) {
                if (kVLo) cb = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlb[ks - KB], fh[ks % NB], cb, 0, 0, 0);
                if (kXLo) cb = __builtin_amdgcn_mfma_f32_32x32x16_f16(whb[ks - KB], fl[ks % NB], cb, 0, 0, 0);
                cb = __builtin_amdgcn_mfma_f32_32x32x16_f16(whb[ks - KB], fh[ks % NB], cb, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (TLDS && SVPS_SHL_PRIO == 1) __builtin_amdgcn_s_setprio(0);
        if constexpr (!TLDS) {
            float ss = sumsq(ca) + sumsq(cb);
            ss += __shfl_xor(ss, 32);
            if (h == 0) part[(cur * 2 + (KEY ? 0 : 1)) * 128 + J * 32 + r] = ss;
        }
    };
    // (TLDS) the sums of squares of `tile`'s finished chains - in the wave's LIGHT half-period, under the other wave's chain
    auto sums = [&](int tile) {
        const int cur = (tile - tile0) % 3;
        float ss = sumsq(ca) + sumsq(cb);
        ss += __shfl_xor(ss, 32);
        if (h == 0) part[(cur * 2 + (KEY ? 0 : 1)) * 128 + J * 32 + r] = ss;
    };
    // (TLDS) the tables of a tile by LDS-DMA into this key wave's own 9 KiB: eight 1-KiB pieces of the tiled Tx' (row block, g: exactly
    // the lanes' 16 bytes in lane order) and the sixteen 16-byte chunks (block, h, g) of the row's Ty'. No registers, no wait: they are
    // requested one period before start_values_lds reads them back (same wave: vmcnt alone orders it).
    const u32x4 srd_tx = shl_make_srd(a.txk, (uint32_t)a.tx_rows * 1024u);
    const u32x4 srd_ty = shl_make_srd(a.tyk, (uint32_t)a.ty_rows * 1024u);
    const int vty = (((lane & 8) ? RBB : RBA) * 32 + (lane & 7) * 4) * 4;
    auto request_tables = [&](int tile) {
        const int xt = tile / rows, y = tile - xt * rows;
        if (tile == tile0 || y == 0) {               // a new 32-pixel column: its Tx' (otherwise the ones in LDS stay)
            const uint32_t dx = __builtin_amdgcn_readfirstlane(lds0 + StatsHlLds::tabx + J * 8192);
            const int sx = __builtin_amdgcn_readfirstlane(xt * 32768);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                shl_dma16(srd_tx, dx + g * 1024, lane * 16, sx + (RBA * 4 + g) * 1024);
                shl_dma16(srd_tx, dx + (4 + g) * 1024, lane * 16, sx + (RBB * 4 + g) * 1024);
            }
        }
        const int sy = __builtin_amdgcn_readfirstlane((y < a.ty_rows ? y : a.ty_rows - 1) * 1024);
        shl_dma16(srd_ty, __builtin_amdgcn_readfirstlane(lds0 + StatsHlLds::taby + J * 1024), vty, sy);
    };
    auto start_values_lds = [&]() {
        if constexpr (KEY) {
            const char* bx = smem + StatsHlLds::tabx + J * 8192 + lane * 16;
            const char* by = smem + StatsHlLds::taby + J * 1024 + h * 64;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 xa = *reinterpret_cast<const f32x4*>(bx + g * 1024), xb = *reinterpret_cast<const f32x4*>(bx + (4 + g) * 1024);
                const f32x4 ya = *reinterpret_cast<const f32x4*>(by + g * 16), yb = *reinterpret_cast<const f32x4*>(by + 128 + g * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ca[4 * g + j] = ya[j] + xa[j];
                    cb[4 * g + j] = yb[j] + xb[j];
                }
            }
        } else {
            const char* bv = smem + StatsHlLds::rbv + h * 64;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 ra_ = *reinterpret_cast<const f32x4*>(bv + RBA * 128 + g * 16), rb_ = *reinterpret_cast<const f32x4*>(bv + RBB * 128 + g * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ca[4 * g + j] = ra_[j];
                    cb[4 * g + j] = rb_[j];
                }
            }
        }
    };

    // PING-PONG (two barriers per tile): in the first half-period the key waves run their chain while the value waves of the same SIMDs
    // do their light work (start values, the DMA requests of tile + 2, the finish of tile - 1); in the second the roles swap (the key
    // waves load the position tables of tile + 1 there). One chain at a time per SIMD: the matrix pipe is never shared.
    // Landing of a tile: a wave's requests for tile + 2 are OLDER than the start-value loads of its next light phase, whose wait
    // (vmcnt retires in order) therefore covers them - one half-period before the first chain reads the tile; the explicit
    // s_waitcnt below states it.
    if constexpr (TLDS) {
        // Round 5 (tiled tables): the start values through LDS too (requested a period ahead), and each wave's sums of squares moved
        // into ITS LIGHT half - a half-period is then the bare chain (54 MFMAs) beside ~1 000 cycles of light work.
        //   half A of tile: key chain(tile)   | value: sums(tile - 1), constants, landing wait, request(tile + 2), finish(tile - 2)
        //   half B of tile: value chain(tile) | key:   sums(tile), landing wait, start values of tile + 1 from LDS, requests of tile + 2
        request(tile0);
        if (tile0 + 1 < tile1) request(tile0 + 1);
        if constexpr (KEY) request_tables(tile0);
        if (threadIdx.x < 256) reinterpret_cast<float*>(smem + StatsHlLds::rbv)[threadIdx.x] = a.rbv[threadIdx.x];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if constexpr (KEY) {
            start_values_lds();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (tile0 + 1 < tile1) request_tables(tile0 + 1);
        }
        if (!KEY && SVPS_SHL_PRIO == 2) __builtin_amdgcn_s_setprio(1);
        for (int tile = tile0; tile < tile1; ++tile) {
            SHL_STAMP(0);
            if constexpr (KEY) {
                heavy(tile);
            } else {
                // tile + 2 is requested at the TOP of the light half, before the landing wait of tile + 1 (its ring slot was last read
                // by the chains of tile - 1): two tiles in flight per CU. With the request after the wait (one tile in flight) the
                // loads took longer than a period to land and both light halves ended in ~1 100 cycles of vmcnt wait.
                const bool ahead = tile + 2 < tile1 && !(SVPS_SHL_ABL & 4);
                if (ahead) request(tile + 2);
                if (tile > tile0 && !(SVPS_SHL_ABL & 32)) sums(tile - 1);
                if (!(SVPS_SHL_ABL & 16)) start_values_lds();
                SHL_STAMP(1);
                // this wave's pieces of tile + 1 (requested a period ago) have landed; the four requests above stay in flight
                // (before the finish: its store counts in vmcnt as well)
                if (ahead) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                SHL_STAMP(2);
                if (J == 0 && tile >= tile0 + 2 && !(SVPS_SHL_ABL & 8)) finish(tile - 2);
            }
            SHL_STAMP(3);
#if !SVPS_SHL_STAGGER
            __syncthreads();
#endif
            SHL_STAMP(4);
            if constexpr (KEY) {
                const bool ahead = tile + 2 < tile1 && !(SVPS_SHL_ABL & 4);
                if (ahead) request(tile + 2);
                if (!(SVPS_SHL_ABL & 32)) sums(tile);
                SHL_STAMP(5);
                if (tile + 1 < tile1) {
                    // the tables of tile + 1 and (older) this wave's pieces of tile + 1 have landed
                    if (ahead) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (!(SVPS_SHL_ABL & 16)) start_values_lds();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // ... and are in registers before the next requests overwrite them
                    if (tile + 2 < tile1 && !(SVPS_SHL_ABL & 64)) request_tables(tile + 2);
                }
                SHL_STAMP(6);
            } else {
                heavy(tile);
                SHL_STAMP(5);
            }
            SHL_STAMP(7);
            __syncthreads();
        }
        if constexpr (!KEY) {
            sums(tile1 - 1);
            if (J == 0 && tile1 - tile0 >= 2 && !(SVPS_SHL_ABL & 8)) finish(tile1 - 2);
        }
        __syncthreads();
        if (!KEY && J == 0 && !(SVPS_SHL_ABL & 8)) finish(tile1 - 1);
        return;
    }
    request(tile0);
    if (tile0 + 1 < tile1) request(tile0 + 1);
    if constexpr (KEY) start_values(tile0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int tile = tile0; tile < tile1; ++tile) {
        SHL_STAMP(0);
        // ---- first half: key heavy, value light
        if constexpr (KEY) {
            heavy(tile);
        } else {
            start_values(tile);
            SHL_STAMP(1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's pieces of tile + 1 have landed (requested a period ago)
            if (tile + 2 < tile1 && !(SVPS_SHL_ABL & 4)) request(tile + 2);
            SHL_STAMP(2);
            if (J == 0 && tile > tile0 && !(SVPS_SHL_ABL & 8)) finish(tile - 1);
        }
        SHL_STAMP(3);
        __syncthreads();
        SHL_STAMP(4);
        // ---- second half: value heavy, key light
        if constexpr (KEY) {
            if (tile + 1 < tile1) start_values(tile + 1);
            SHL_STAMP(5);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's pieces of tile + 1 have landed
            if (tile + 2 < tile1 && !(SVPS_SHL_ABL & 4)) request(tile + 2);
            SHL_STAMP(6);
        } else {
            heavy(tile);
            SHL_STAMP(5);
        }
        SHL_STAMP(7);
        __syncthreads();                             // tile + 1 is complete in LDS; the sums of `tile` are complete
    }
    if (!KEY && J == 0 && !(SVPS_SHL_ABL & 8)) finish(tile1 - 1);
}

template <bool TLDS>
__global__ __launch_bounds__(512) void retr_stats_hl_kernel(StatsHlArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    // waves w and w + 4 share a SIMD: the key and the value wave of one quarter - while one waits (tables, tile loads, sums) the other
    // has MFMAs to issue
    switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {
        case 0: stats_hl_role<0, true, TLDS>(a, smem, lane); break;
        case 1: stats_hl_role<1, true, TLDS>(a, smem, lane); break;
        case 2: stats_hl_role<2, true, TLDS>(a, smem, lane); break;
        case 3: stats_hl_role<3, true, TLDS>(a, smem, lane); break;
        case 4: stats_hl_role<0, false, TLDS>(a, smem, lane); break;
        case 5: stats_hl_role<1, false, TLDS>(a, smem, lane); break;
        case 6: stats_hl_role<2, false, TLDS>(a, smem, lane); break;
        default: stats_hl_role<3, false, TLDS>(a, smem, lane); break;
    }
}

}  // namespace svps

// svps_retr_stats_hl_fwd (include/slotvps_hip.h): tyk [ty_rows, 256] = Ty + r_k, txk [tx_rows, 256] = Tx, rbv [256] - fp32, columns in
// ACCUMULATOR order (column 32 B + 16 h + 4 g + j = factor row 32 B + 8 g + 4 h + j); ty_rows = H or 1, tx_rows = W or 1 (no position term:
// one row each, txk zero). tx_tiled (W % 32 == 0): txk re-ordered to [W / 32][8 B][4 g][2 h][32 pixels][4 j].
extern "C" int svps_retr_stats_hl_fwd(const void* feat_hi, const void* feat_lo, const float* tyk, int ty_rows, const float* txk, int tx_rows, int tx_tiled,
                                      const void* rk_hi, const void* rk_lo, float lnk_eps, const void* rv_hi, const void* rv_lo,
                                      const float* rbv, float lnv_eps, void* aux, int T, int H, int W, int D, void* stream_) {
    if (!feat_hi || !feat_lo || !tyk || !txk || !rk_hi || !rk_lo || !rv_hi || !rv_lo || !rbv || !aux) return SVPS_ERR_BAD_ARG;
    if (D != svps::kD || T <= 0 || H <= 0 || W <= 0) return SVPS_ERR_BAD_SHAPE;
    if ((ty_rows != H && ty_rows != 1) || (tx_rows != W && tx_rows != 1)) return SVPS_ERR_BAD_SHAPE;
    if (tx_tiled && (tx_rows != W || (W & 31))) return SVPS_ERR_BAD_SHAPE;
    if ((size_t)H * W > svps::kMaxFramePixels) return SVPS_ERR_BAD_SHAPE;
    const int HW = H * W;
    const int tiles = (HW + svps::kTilePx - 1) / svps::kTilePx;
    int chunks = svps_pick_chunks(T, tiles, svps_num_cus());
    const int tpw = (tiles + chunks - 1) / chunks;
    chunks = (tiles + tpw - 1) / tpw;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    using H16 = _Float16;
    static const bool no_tlds = [] { const char* e = getenv("SVPS_SHL_TLDS"); return e && e[0] == '0'; }();
    const bool tlds = tx_tiled && ty_rows == H && !no_tlds;
    static SvpsLdsAttr attr, attr_t;
    if (hipError_t ae = tlds ? attr_t.ensure(reinterpret_cast<const void*>(svps::retr_stats_hl_kernel<true>), svps::StatsHlLds::total_tables)
                             : attr.ensure(reinterpret_cast<const void*>(svps::retr_stats_hl_kernel<false>), svps::StatsHlLds::total);
        ae != hipSuccess)
        return (int)ae;
    const svps::StatsHlArgs args{static_cast<const H16*>(feat_hi), static_cast<const H16*>(feat_lo), static_cast<const H16*>(rk_hi),
                                 static_cast<const H16*>(rk_lo), static_cast<const H16*>(rv_hi), static_cast<const H16*>(rv_lo), tyk, txk, rbv,
                                 static_cast<H16*>(aux), lnk_eps, lnv_eps, HW, W, ty_rows, tx_rows, tpw, tx_tiled ? 1 : 0};
    svps_prof_mark(SVPS_KERNEL_RETR_STATS, 0, stream);
    if (tlds)
        hipLaunchKernelGGL(svps::retr_stats_hl_kernel<true>, dim3(chunks, T), dim3(512), svps::StatsHlLds::total_tables, stream, args);
    else
        hipLaunchKernelGGL(svps::retr_stats_hl_kernel<false>, dim3(chunks, T), dim3(512), svps::StatsHlLds::total, stream, args);
    svps_prof_mark(SVPS_KERNEL_RETR_STATS, 1, stream);
    return (int)hipGetLastError();
}

#ifdef SVPS_SHL_STAMP
extern "C" int svps_shl_debug_read(unsigned long long* stamps) {
    return (int)hipMemcpyFromSymbol(stamps, HIP_SYMBOL(svps::shl_stamps), sizeof(unsigned long long) * 2 * 8 * 8);
}
#endif
