"""(GPU box) K6c alone: the clip post-process kernels (csrc/panoptic_clip.hip) on synthetic kept-slot logits, per stage, with warmed clocks.
usage: python tools/kbench_ppc.py [--T 5] [--K 20] [--h 256] [--w 512] [--things 12]"""
import argparse, ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slotvps_amd import _lib
from slotvps_amd import postprocess as P

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=5)
ap.add_argument("--K", type=int, default=20)
ap.add_argument("--h", type=int, default=256)
ap.add_argument("--w", type=int, default=512)
ap.add_argument("--things", type=int, default=12)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
lib = _lib.load()
T, K, h, w = a.T, a.K, a.h, a.w
H, W = 4 * h, 4 * w
g = torch.Generator(device="cpu").manual_seed(0)
yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
m = torch.empty(T, K, h, w)
for t in range(T):
    for k in range(K):
        cy, cx, s = torch.rand(1, generator=g) * h, torch.rand(1, generator=g) * w, 10 + torch.rand(1, generator=g) * 0.25 * w
        m[t, k] = 9.0 * torch.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s)) - 3.0 + 0.1 * torch.randn(h, w, generator=g)
m = m.to(dev).contiguous()
st_h = np.zeros((T, P.PPC_STATE_INTS), dtype=np.int32)
st_h[:, P.PPC_K] = K
st_h[:, P.PPC_THING + K - a.things:P.PPC_THING + K] = 1
st_h[:, P.PPC_CL:P.PPC_CL + K] = np.arange(K) % 19
st0 = torch.from_numpy(st_h).to(dev)
state = st0.clone()
pairs = torch.zeros((T, K * K), dtype=torch.int32, device=dev)
cand = torch.empty((T, H * W, 2), dtype=torch.uint8, device=dev)
ids = torch.empty((T, H, W), dtype=torch.uint8, device=dev)
vp = lambda t: ctypes.c_void_p(t.data_ptr())
s = torch.cuda.current_stream().cuda_stream


def call(rounds, stages):
    rc = lib.svps_panoptic_clip(vp(m), K * h * w, T, h, w, H, W, vp(state), vp(pairs), K * K, vp(cand), vp(ids), 0.4, 0.03, 0, 11, rounds, stages, s)
    assert rc == 0, rc


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


x = torch.randn(4096, 4096, device=dev)
for _ in range(20):
    x @ x                                   # clocks up
torch.cuda.synchronize()


def cand_only():
    state.copy_(st0); pairs.zero_()
    call(0, 1)


def one_round():
    call(1, 2)


def whole():
    state.copy_(st0); pairs.zero_()
    call(4, 7)


def reset():
    state.copy_(st0); pairs.zero_()


t_reset = timed(reset, a.reps)
t_cand = timed(cand_only, a.reps) - t_reset
cand_only(); torch.cuda.synchronize()
n_before = state[:, P.PPC_N].tolist()
# an area round on frames in phase 0 (state is put back each time: the step kernel would finish them)
snap = state.clone()


def area_round():
    state.copy_(snap)
    call(1, 2)


t_round = timed(area_round, a.reps) - timed(lambda: state.copy_(snap), a.reps)
t_all = timed(whole, a.reps) - t_reset
whole(); torch.cuda.synchronize()
print(f"T={T} K={K} ({a.things} things) {h}x{w} -> {H}x{W}: candidates + decide {t_cand:.1f} us, one (area pass + step) {t_round:.1f} us, "
      f"whole sequence (4 rounds enqueued) {t_all:.1f} us; kept per frame {n_before}, final {state[:, P.PPC_N].tolist()}, rounds {state[:, P.PPC_ROUNDS].tolist()}, "
      f"phase {state[:, P.PPC_PHASE].tolist()}")
