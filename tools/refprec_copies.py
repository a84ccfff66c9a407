"""(GPU box) Which host ops of one eager fp16x2 step launch device-to-device copies (hipMemcpyAsync: `Memcpy DtoD` in the profile)
and the framework's element-wise kernels: torch.profiler with stacks, grouped by the innermost frame inside slotvps_amd/."""
import collections, os, sys
import torch
from torch.profiler import profile, ProfilerActivity
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slotvps_amd.clip import SlotClipRunner

dev = torch.device("cuda:0")
r = SlotClipRunner(dev, 5, 1024, 2048, L=100, use_graph=False, clips_per_launch=int(os.environ.get("CPL", "4")))
r.head.set_mode(os.environ.get("PRECISION", "fp16x2"))
r.load_clip(r.random_clip(1))
for _ in range(2):
    r.run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    r.run()
    torch.cuda.synchronize()
by = collections.Counter()
dur = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::contiguous", "aten::clone", "aten::cat", "aten::stack", "aten::gelu", "aten::relu_", "aten::add", "aten::mul",
                   "aten::softmax", "aten::_softmax", "aten::bmm", "aten::matmul", "aten::linear", "aten::addmm", "aten::zeros", "aten::fill_", "aten::zero_",
                   "aten::to", "aten::_to_copy", "aten::clamp", "aten::clamp_", "aten::sum", "aten::any", "aten::abs", "aten::max"):
        frames = [f for f in (ev.stack or []) if "slotvps_amd" in f]
        where = frames[0].split("slotvps_amd/")[-1] if frames else "?"
        by[(ev.name, where)] += 1
        dur[(ev.name, where)] += ev.device_time_total
for (name, where), n in sorted(by.items(), key=lambda kv: -dur[kv[0]])[:40]:
    print(f"{n:4d} x {name:18s} {dur[(name, where)] / 1e3:8.3f} ms device  {where}")
