import math, sys, time, torch
sys.path.insert(0, "/root/repo")
import torch.nn.functional as F
from slotvps_amd import ops
dev = torch.device("cuda:0")
T, L, C, nh = 80, 100, 256, 8
qkv = torch.randn(T, L, 3 * C, device=dev)
def a():
    return ops.slot_self_attn(qkv, nh)
def b():
    v5 = qkv.view(T, L, 3, nh, C // nh)
    q, k, v = (v5[:, :, i].transpose(1, 2) for i in range(3))
    attn = torch.softmax((q * (1.0 / math.sqrt(C // nh))) @ k.transpose(-1, -2), dim=-1)
    return (attn @ v).transpose(1, 2).reshape(T, L, C)
def c():
    v5 = qkv.view(T, L, 3, nh, C // nh)
    q, k, v = (v5[:, :, i].transpose(1, 2) for i in range(3))
    return F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(T, L, C)
for name, fn in (("hip kernel", a), ("matmul+softmax", b), ("sdpa", c)):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): o = fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us  err vs hip {(o - a()).abs().max().item():.2e}")
