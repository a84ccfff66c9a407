/*
 * slotvps_hip_diag.h - C ABI of the DIAGNOSTICS library of the MI355X slot-retriever path (libslotvps_hip_diag.so; round 6).
 *
 * Not part of the product: nothing in slotvps_amd/ needs it to run. It holds what measures and pins the product library
 * (include/slotvps_hip.h, libslotvps_hip.so) from outside:
 *   - the per-kernel device-time accounting of bench.py's roofline leg: HIP events on the launch stream, driven through the product's
 *     launch hook (svps_set_launch_hook): svps_diag_launch_hook is the callback to install
 *   - hardware-semantics probes (tests/test_probes_gpu.py): MFMA 32x32x16 operand / result lane maps, ds_read_b64_tr_b16, the swizzled
 *     LDS-DMA tile image - the instruction behaviour the kernels rely on
 *   - streaming probes: the known-bytes copy the HBM counters are calibrated on (tools/pmc_traffic.py), read : write mixes, the MFMA
 *     feed / sustain probes (tools/mfma_feed_probe.py, tools/mfma_sustain_probe.py)
 * Same calling rules as the product ABI: device pointers, stream as void*, 0 or an error code, no allocation of device memory.
 */
#ifndef SLOTVPS_HIP_DIAG_H_
#define SLOTVPS_HIP_DIAG_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- per-kernel device time: when enabled, every launch the product brackets with svps_prof_mark is bracketed by HIP events on the
 * launch stream. svps_prof_collect synchronises those events (host-blocking) and returns the summed device time of one kernel id
 * (SVPS_KERNEL_* of slotvps_hip.h). Install with svps_set_launch_hook(svps_diag_launch_hook) of the product library. */
void svps_diag_launch_hook(int kernel_id, int is_end, void* stream);
void svps_prof_enable(int on);
void svps_prof_reset(void);
int svps_prof_collect(int kernel_id, double* total_ms, int* launches);

/* ---- hardware-semantics probes
 *   probe_mfma: a [32,16] bf16, b [16,32] bf16 (row-major) -> c [32,32] fp32 = a @ b
 *   probe_tile: x [32, 256] bf16 -> rows [32,256] (through read_row_frag) and cols [32,256] (through read_col_frag), both must reproduce x */
int svps_probe_mfma(const void* a, const void* b, float* c, void* stream);
int svps_probe_tile(const void* x, void* rows, void* cols, void* stream);
/* svps_probe_copy: dst[0:bytes] = src[0:bytes] with 16 B per lane streaming loads / stores (bytes a multiple of 16): the
 * known-bytes kernel the HBM counters are calibrated on and the hand-written copy ceiling of bench.py */
int svps_probe_copy(const void* src, void* dst, size_t bytes, void* stream);
/* svps_probe_mix: mixed-traffic streaming probe - per unit ri KiB are read from src and ro KiB written to dst (src >= units * ri
 * KiB, dst >= units * ro KiB, at least 1 KiB), every byte once: the read : write mix of a kernel without its arithmetic, to state
 * the box's ceiling for that mix (K4: 5 : 4). (ri, ro) in {(5,4), (3,4), (1,1), (1,0), (0,1), (4,1), (2,1)}. */
int svps_probe_mix(const void* src, void* dst, size_t units, int ri, int ro, void* stream);
/* svps_probe_mfma_feed: cycles per MFMA of a dependent / independent v_mfma_f32_32x32x16_f16 stream under several operand feeds
 * (tools/mfma_feed_probe.py, tools/mfma_sustain_probe.py) */
int svps_probe_mfma_feed(int mode, int tiles, int nact, int blocks, unsigned long long* out_dev, float* sink_dev, void* stream);

/* svps_probe_mx_fp8: one v_mfma_scale_f32_32x32x64_f8f6f4 (both operands FP8 e4m3) per problem on RAW operand registers - a_regs / b_regs
 * [n][64 lanes][8 dwords], scale_a / scale_b [n][64] (the scale VGPRs), c_out [n][64][16] fp32 from a zero accumulator: pins the
 * instruction's lane maps and scale semantics (tools/mx_probe.py) */
/* svps_probe_cvt_fp8: out[i] = result dword of v_cvt_scalef32_pk_fp8_f16 on the fp16 pair x_pairs[i] with `scale` (bytes 0, 1) */
int svps_probe_cvt_fp8(const void* x_pairs, float scale, int* out, int n, void* stream);
/* svps_probe_mx_block: a [32][64], b [32][64] fp16 -> c [64 lanes][16] = a b^T through the in-kernel FP8 conversion (four fp16 k-step
 * fragments per lane, scale bytes sa / sb [64]) and ONE scaled MFMA - the building block of retr_stats_hl.hip's F8 form; regs_out [64][16]:
 * the operand registers as converted */
int svps_probe_mx_block(const void* a, const void* b, const void* sa, const void* sb, float* c_out, int* regs_out, void* stream);
int svps_probe_mx_fp8(const void* a_regs, const void* b_regs, const void* scale_a, const void* scale_b, float* c_out, int n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SLOTVPS_HIP_DIAG_H_ */
