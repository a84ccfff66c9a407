"""A few K1 launches at the headline level (for rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
ln = torch.nn.functional.layer_norm
T, HW = 5, 256 * 512
q = ln(torch.randn((T, 100, 256), generator=g, device=dev), (256,)).to(torch.bfloat16)
k = ln(torch.randn((T, HW, 256), generator=g, device=dev), (256,)).to(torch.bfloat16)
v = ln(torch.randn((T, HW, 256), generator=g, device=dev), (256,)).to(torch.bfloat16)
w = torch.ones(256, device=dev); b = torch.zeros(256, device=dev)
for _ in range(5):
    ops.slot_attn(q, k, v, w, b, split_p=True)
torch.cuda.synchronize()
print("done", 2 * T * HW * 512 / 1e6, "MB algorithmic per launch")
