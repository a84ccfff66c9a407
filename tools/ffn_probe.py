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

print("chains of 256 -> 256 + LayerNorm layers (csrc/slot_chain.hip) against one K8 launch per layer:")
for M in (16000, 8000, 500):
    x = torch.randn((M, 256), generator=g, device=dev)
    W = [ops.pack_b_fragments(torch.randn((256, 256), generator=g, device=dev) / 16.0) for _ in range(4)]
    G = [torch.ones(256, device=dev) for _ in range(4)]
    E = [torch.zeros(256, device=dev) for _ in range(4)]

    def tower_k8():
        r = x
        for i in range(3):
            r = ops.slot_gemm_ln(r, W[i], None, G[i], E[i], 1e-5, relu=True)
        return r, ops.slot_gemm_ln(x, W[3], None, G[3], E[3], 1e-5, relu=True)
    tower = [dict(wpack=W[0], gamma=G[0], beta=E[0], relu=True), dict(wpack=W[1], gamma=G[1], beta=E[1], relu=True),
             dict(wpack=W[2], gamma=G[2], beta=E[2], relu=True, out=True), dict(wpack=W[3], gamma=G[3], beta=E[3], relu=True, src="x")]
    qkv = [dict(wpack=W[i], gamma=G[i], beta=E[i], src="x", out=True) for i in range(3)]
    t_a, _ = timed_graph(tower_k8)
    t_b, _ = timed_graph(lambda: ops.slot_chain(x, tower))
    t_c, _ = timed_graph(lambda: [ops.slot_gemm_ln(x, W[i], None, G[i], E[i], 1e-5) for i in range(3)])
    t_d, _ = timed_graph(lambda: ops.slot_chain(x, qkv))
    print(f"M={M}: tower 4 launches {t_a:6.1f} us, chain {t_b:6.1f} us; three projections 3 launches {t_c:6.1f} us, chain {t_d:6.1f} us")
