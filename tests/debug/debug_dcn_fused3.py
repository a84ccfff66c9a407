import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from slotvps_amd.dcn import DeformConv
cuda = torch.device("cuda:0")
C, O, H, W, scale = 256, 256, 16, 32, 2.5
torch.manual_seed(0)
m = DeformConv(C, O, 3, padding=1).to(cuda)
x = torch.randn(1, C, H, W, device=cuda)
off = scale * torch.randn(1, 18, H, W, device=cuda)
with torch.no_grad():
    m.fused = False; b = m(x, off); m.fused = True
    outs = [m(x, off).clone() for _ in range(6)]
torch.cuda.synchronize()
print("env", os.environ.get("SVPS_DCN_NOPREFETCH"), "errs vs im2col:", [f"{(o - b).abs().max().item():.1e}" for o in outs],
      "bitwise equal runs:", [bool(torch.equal(outs[0], o)) for o in outs[1:]])
