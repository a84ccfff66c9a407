"""The feed-forward block of a stage: one launch (csrc/slot_ffn.hip) against the two K8 launches, device time through a hipGraph."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd import ops
from tools.gemm_probe_util import timed_graph

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for (M, H, act) in [(16000, 2048, ops.ACT_GELU), (16000, 1024, ops.ACT_RELU), (500, 2048, ops.ACT_GELU), (8000, 2048, ops.ACT_GELU)]:
    x = torch.randn((M, 256), generator=g, device=dev)
    w1 = torch.randn((H, 256), generator=g, device=dev) / 16.0
    b1 = 0.1 * torch.randn((H,), generator=g, device=dev)
    w2 = torch.randn((256, H), generator=g, device=dev) / H ** 0.5
    b2 = 0.1 * torch.randn((256,), generator=g, device=dev)
    gamma = torch.ones(256, device=dev)
    beta = torch.zeros(256, device=dev)
    p1, p2 = ops.pack_b_fragments(w1), ops.pack_b_fragments(w2)
    t2, _ = timed_graph(lambda: ops.slot_gemm_ln(ops.slot_gemm(x, p1, b1, act), p2, b2, gamma, beta, 1e-5, pre=x))
    t1, _ = timed_graph(lambda: ops.slot_ffn(x, p1, b1, p2, b2, gamma, beta, 1e-5, act=act, pre=x))
    fl = 3 * 2 * 2 * M * 256 * H
    print(f"M={M} H={H}: two launches {t2:7.1f} us, one launch {t1:7.1f} us ({fl / t1 / 1e6:.0f} TFLOP/s executed)")
