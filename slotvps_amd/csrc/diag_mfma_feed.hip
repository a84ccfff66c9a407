// Diagnostic probe, compiled only into the stamp library (make stamp): how fast can ONE wave run a 32-MFMA chain with its B operand
// from LDS / registers, with vector fillers between the MFMAs, beside a second wave on the SIMD? tools/mfma_feed_probe.py.
#include "common.h"
#include "../../include/slotvps_hip.h"
#include "../../include/slotvps_hip_diag.h"

namespace svps {
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
}

// ---- diagnostic build only: how fast can ONE wave run the producer's chain (32 MFMA 32x32x16 per tile, B operand from LDS)?
// tools/mfma_feed_probe.py. MODE bit 0-1: 0 = B fragments from LDS in double-buffered groups of four, one accumulator (the
// producer's loop); 1 = B fragments in registers (no LDS traffic); 2 = from LDS, two accumulators; 3 = from LDS, ring of three
// groups (two groups ahead). bit 2 (4): a workgroup barrier per tile. `nact` waves of the 8 run the loop (4 = one per SIMD).
namespace svps {
template <int MODE>
__global__ __launch_bounds__(512) void mfma_feed_kernel(unsigned long long* __restrict__ out, float* __restrict__ sink, int tiles, int nact) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>((SVPS_LDS const void*)smem);
    {   // two tiles of pseudo-random fp16 in [-1, 1)
        uint32_t x = 0x9e3779b9u * (threadIdx.x + 1) + blockIdx.x;
        for (int i = threadIdx.x; i < 2 * kTileBytes / 4; i += 512) {
            x = x * 1664525u + 1013904223u;
            const _Float16 a = (_Float16)((float)(int)(x >> 16 & 0xffff) * (1.f / 32768.f) - 1.f);
            const _Float16 b = (_Float16)((float)(int)(x & 0xffff) * (1.f / 32768.f) - 1.f);
            reinterpret_cast<uint32_t*>(smem)[i] = (uint32_t)__builtin_bit_cast(uint16_t, a) | ((uint32_t)__builtin_bit_cast(uint16_t, b) << 16);
        }
    }
    f16x8 qfh[16], qfl[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            qfh[ks][j] = (_Float16)(0.01f * (float)((lane * 7 + ks * 3 + j) % 17 - 8));
            qfl[ks][j] = (_Float16)(0.001f * (float)((lane * 5 + ks + j) % 13 - 6));
        }
    __syncthreads();
    const uint32_t lane_row = lds0 + r * kRowBytes + ((h ^ swz(r)) << 4);
    auto frag = [&](uint32_t tb, int ks) {
        return *reinterpret_cast<SVPS_LDS const f16x8*>((uintptr_t)((tb ^ ((ks & 7) << 5)) + 256 * (ks >> 3)));
    };
    constexpr int kOrd[4] = {0, 8, 1, 9};
    constexpr int M = MODE & 3;
    constexpr bool BAR = (MODE & 4) != 0;
    constexpr int FILL = ((MODE >> 4) & 3) * 2;                  // independent v_fma_f32 after EVERY MFMA (0, 2, 4, 6)
    constexpr int FKIND = (MODE >> 6) & 3;                       // 1: FILL / 2 v_pk_add_f32 instead; 2: FILL v_exp_f32 instead
    float fv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) fv[i] = (float)(lane + i);
    float fc1 = 1.0001f, fc2 = 0.25f;
    asm volatile("" : "+v"(fc1), "+v"(fc2));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    f32x2_t fp[4], fpc = {0.5f, 0.25f};
#pragma unroll
    for (int i = 0; i < 4; ++i) { fp[i][0] = (float)lane; fp[i][1] = (float)i; }
    asm volatile("" : "+v"(fpc));
    f32x16 s, s2;
#pragma unroll
    for (int i = 0; i < 16; ++i) { s[i] = 0.f; s2[i] = 0.f; }
    f16x8 kreg[16];
    if constexpr (M == 1) {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) kreg[ks] = frag(lane_row, ks);
    }
    unsigned long long t0 = 0;
    if (lane == 0) t0 = __builtin_amdgcn_s_memtime();
    if (nact < 0 && w >= 4) {
        // probe of the arbitration: the younger half runs vector arithmetic / LDS reads only (256 dependent-free FMAs + 8 reads per "tile")
        // while the older half (nact == -1) runs the MFMA chain, or alone (nact == -2)
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = (float)(lane + i);
        for (int it = 0; it < tiles; ++it) {
#pragma unroll
            for (int rep = 0; rep < 16; ++rep)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(fc1), "v"(fc2));
            const uint32_t tb = lane_row + (uint32_t)(it & 1) * kTileBytes;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) { const f16x8 f = frag(tb, ks); v[ks] += (float)f[0]; }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] += v[i];
    } else if ((nact == -2) && w < 4) {
        // older half idle
    } else if (w < (nact < 0 ? 4 : nact) || BAR) {
        if (nact < 0) nact = 4;
        for (int it = 0; it < tiles; ++it) {
            if constexpr (BAR) wg_barrier();
            if (w >= nact) continue;
            const uint32_t tb = lane_row + (uint32_t)(it & 1) * kTileBytes;
            if constexpr (M == 1) {
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qfh[ks], kreg[ks], s, 0, 0, 0);
                    if constexpr (FILL > 0) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < FILL; ++i) {
                            if constexpr (FKIND == 2) asm volatile("v_exp_f32 %0, %1" : "=v"(fv[i]) : "v"(fc1));
                            else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fv[i]) : "v"(fc1), "v"(fc2));
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qfl[ks], kreg[ks], s, 0, 0, 0);
                    if constexpr (FILL > 0) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < FILL; ++i) {
                            if constexpr (FKIND == 2) asm volatile("v_exp_f32 %0, %1" : "=v"(fv[(i + FILL) & 7]) : "v"(fc2));
                            else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fv[(i + FILL) & 7]) : "v"(fc2), "v"(fc1));
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else if constexpr (M == 3) {
                f16x8 kf[3][4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { kf[0][u] = frag(tb, kOrd[u]); kf[1][u] = frag(tb, 2 + kOrd[u]); }
#pragma unroll
                for (int grp = 0; grp < 4; ++grp) {
                    if (grp < 2) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) kf[(grp + 2) % 3][u] = frag(tb, 2 * (grp + 2) + kOrd[u]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qfh[2 * grp + kOrd[u]], kf[grp % 3][u], s, 0, 0, 0);
                        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qfl[2 * grp + kOrd[u]], kf[grp % 3][u], s, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                f16x8 kf[2][4];
#pragma unroll
                for (int u = 0; u < 4; ++u) kf[0][u] = frag(tb, kOrd[u]);
#pragma unroll
                for (int grp = 0; grp < 4; ++grp) {
                    if (grp < 3) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) kf[(grp + 1) & 1][u] = frag(tb, 2 * (grp + 1) + kOrd[u]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qfh[2 * grp + kOrd[u]], kf[grp & 1][u], s, 0, 0, 0);
                        if constexpr (FILL > 0) {
                            __builtin_amdgcn_sched_barrier(0);
                            if constexpr (FKIND == 1) { for (int i = 0; i < FILL / 2; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(fp[i]) : "v"(fpc)); }
                            else if constexpr (FKIND == 2) { for (int i = 0; i < FILL; ++i) asm volatile("v_exp_f32 %0, %1" : "=v"(fv[i]) : "v"(fc1)); }
                            else for (int i = 0; i < FILL; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fv[i]) : "v"(fc1), "v"(fc2));
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if constexpr (M == 2) s2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(qfl[2 * grp + kOrd[u]], kf[grp & 1][u], s2, 0, 0, 0);
                        else s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qfl[2 * grp + kOrd[u]], kf[grp & 1][u], s, 0, 0, 0);
                        if constexpr (FILL > 0) {
                            __builtin_amdgcn_sched_barrier(0);
                            if constexpr (FKIND == 1) { for (int i = 0; i < FILL / 2; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(fp[(i + FILL / 2) & 3]) : "v"(fpc)); }
                            else if constexpr (FKIND == 2) { for (int i = 0; i < FILL; ++i) asm volatile("v_exp_f32 %0, %1" : "=v"(fv[(i + FILL) & 7]) : "v"(fc2)); }
                            else for (int i = 0; i < FILL; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fv[(i + FILL) & 7]) : "v"(fc2), "v"(fc1));
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += s[i] + s2[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += fv[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc += fp[i][0] + fp[i][1];
    if (lane == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        out[(blockIdx.x * 8 + w) * 2] = t0;
        out[(blockIdx.x * 8 + w) * 2 + 1] = t1;
    }
    if (acc == 12345.678f) sink[threadIdx.x] = acc;
}
}  // namespace svps

extern "C" int svps_probe_mfma_feed(int mode, int tiles, int nact, int blocks, unsigned long long* out_dev, float* sink_dev, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int lds = 2 * svps::kTileBytes;
#define SVPS_FEED(MD) case MD: hipLaunchKernelGGL(svps::mfma_feed_kernel<MD>, dim3(blocks), dim3(512), lds, stream, out_dev, sink_dev, tiles, nact); break;
    switch (mode) {
        SVPS_FEED(0) SVPS_FEED(1) SVPS_FEED(2) SVPS_FEED(3) SVPS_FEED(4) SVPS_FEED(5) SVPS_FEED(6) SVPS_FEED(7)
        SVPS_FEED(16) SVPS_FEED(18) SVPS_FEED(32) SVPS_FEED(34) SVPS_FEED(48) SVPS_FEED(50)
        SVPS_FEED(80) SVPS_FEED(96) SVPS_FEED(112) SVPS_FEED(144) SVPS_FEED(160) SVPS_FEED(17) SVPS_FEED(33) SVPS_FEED(49) SVPS_FEED(145)
        default: return SVPS_ERR_BAD_ARG;
    }
#undef SVPS_FEED
    return (int)hipGetLastError();
}
