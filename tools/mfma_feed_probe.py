"""How fast does one wave run a chain of 32 v_mfma_f32_32x32x16_f16 per 32-pixel tile when the B operand comes from LDS (the
producer loop of K1', the heavy phase of K3')? The probe lives in the diagnostics library
(libslotvps_hip_diag.so, built by `make -C slotvps_amd/csrc`): python tools/mfma_feed_probe.py"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd import _lib, ops

dev = torch.device("cuda:0")
lib = _lib.load_diag()          # the probes live in the diagnostics library (libslotvps_hip_diag.so)
lib.svps_probe_mfma_feed.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
blocks, tiles = 256, 400
out = torch.zeros((blocks * 8, 2), dtype=torch.int64, device=dev)
sink = torch.zeros(512, dtype=torch.float32, device=dev)
names = {0: "B from LDS, groups of 4 double-buffered, one accumulator", 1: "B in registers (no LDS)", 2: "B from LDS, two accumulators",
         3: "B from LDS, ring of three groups"}
for bar in (0, 4):
    for mode in (1, 0, 2, 3):
        for nact in (4, 8):
            for rep in range(3):
                out.zero_()
                rc = lib.svps_probe_mfma_feed(mode | bar, tiles, nact, blocks, out.data_ptr(), sink.data_ptr(), ops._stream_ptr(dev))
                assert rc == 0, rc
                torch.cuda.synchronize()
            o = out.cpu().numpy().reshape(blocks, 8, 2)
            d = (o[:, 0, 1] - o[:, 0, 0]).astype(np.float64)
            d4 = (o[:, 4, 1] - o[:, 4, 0]).astype(np.float64)
            span = (o[:, :nact, 1].max(axis=1) - o[:, :nact, 0].min(axis=1)).astype(np.float64)
            print(f"{names[mode]:58s} barrier/tile={'yes' if bar else 'no '} waves/SIMD={nact // 4}: "
                  f"{np.median(d) / (tiles * 32):6.1f} cycles per MFMA (wave 0), wave 4 {np.median(d4) / (tiles * 32):6.1f}, "
                  f"workgroup span {np.median(span) / (tiles * 32):6.1f}", flush=True)

# arbitration between the two waves of a SIMD: the younger half runs 256 FMAs + 8 LDS reads per tile, alone and beside the older
# half's MFMA chain
for nact, what in ((-2, "younger half alone (vector + LDS work)"), (-1, "younger half beside the older half's MFMA chain")):
    for rep in range(3):
        out.zero_()
        rc = lib.svps_probe_mfma_feed(0, tiles, nact, blocks, out.data_ptr(), sink.data_ptr(), ops._stream_ptr(dev))
        assert rc == 0, rc
        torch.cuda.synchronize()
    o = out.cpu().numpy().reshape(blocks, 8, 2)
    d0 = (o[:, 0, 1] - o[:, 0, 0]).astype(np.float64)
    d4 = (o[:, 4, 1] - o[:, 4, 0]).astype(np.float64)
    print(f"{what:52s}: wave 4 {np.median(d4) / tiles:7.0f} cycles per tile, wave 0 {np.median(d0) / tiles:7.0f}", flush=True)

# independent vector instructions (asm, pinned in place) between the MFMAs of ONE wave's chain, one wave per SIMD: hidden in the
# matrix shadow or not?
def run(mode, nact=4):
    for rep in range(3):
        out.zero_()
        rc = lib.svps_probe_mfma_feed(mode, tiles, nact, blocks, out.data_ptr(), sink.data_ptr(), ops._stream_ptr(dev))
        assert rc == 0, rc
        torch.cuda.synchronize()
    o = out.cpu().numpy().reshape(blocks, 8, 2)
    return np.median((o[:, 0, 1] - o[:, 0, 0]).astype(np.float64)) / (tiles * 32)

for fill in (1, 2, 3):
    print(f"{2 * fill} v_fma_f32 after every MFMA: B in registers {run(1 | fill << 4):6.1f}, B from LDS one accumulator {run(fill << 4):6.1f}, "
          f"two accumulators {run(2 | fill << 4):6.1f} cycles per MFMA", flush=True)
for fill in (1, 2, 3):
    print(f"{fill} v_pk_add_f32 after every MFMA (B from LDS): {run(64 | fill << 4):6.1f} cycles per MFMA", flush=True)
print(f"2 v_exp_f32 after every MFMA: B in registers {run(128 | 16 | 1):6.1f}, B from LDS {run(128 | 16):6.1f}; 4 v_exp_f32, B from LDS: {run(128 | 32):6.1f}", flush=True)
