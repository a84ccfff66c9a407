"""Sustained matrix-pipe rate: the register-only MFMA chain of tools/mfma_feed_probe.py run for ~20 ms per launch, back to back."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd import _lib, ops
dev = torch.device("cuda:0")
lib = _lib.load_diag()          # the probes live in the diagnostics library (libslotvps_hip_diag.so)
lib.svps_probe_mfma_feed.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
blocks = 256
out = torch.zeros((blocks * 8, 2), dtype=torch.int64, device=dev)
sink = torch.zeros(512, dtype=torch.float32, device=dev)
for mode, nact, name in ((1, 4, "B in registers, 1 wave/SIMD"), (1, 8, "B in registers, 2 waves/SIMD"), (0, 4, "B from LDS, 1 wave/SIMD"), (0, 8, "B from LDS, 2 waves/SIMD")):
    for tiles in (400, 40000):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for rep in range(2):
            lib.svps_probe_mfma_feed(mode, tiles, nact, blocks, out.data_ptr(), sink.data_ptr(), ops._stream_ptr(dev))
        torch.cuda.synchronize()
        n = 10 if tiles > 1000 else 50
        ev[0].record()
        for rep in range(n):
            rc = lib.svps_probe_mfma_feed(mode, tiles, nact, blocks, out.data_ptr(), sink.data_ptr(), ops._stream_ptr(dev))
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / n
        o = out.cpu().numpy().reshape(blocks, 8, 2)
        d = (o[:, 0, 1] - o[:, 0, 0]).astype(np.float64)
        cyc = np.median(d) / (tiles * 32)
        mf = blocks * nact * tiles * 32
        print(f"{name:32s} tiles={tiles:6d}: {ms:8.3f} ms per launch, {cyc:5.1f} cycles per MFMA (wave 0), {mf * 32768 / ms / 1e9:7.1f} TFLOP/s, "
              f"clock ~ {np.median(d) / (ms * 1e3):6.0f} MHz (if the launch is all chain)", flush=True)
