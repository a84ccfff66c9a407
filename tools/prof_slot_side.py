"""torch.profiler view of one eager clip step: which host ops launch the small slot-side kernels."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd.clip import SlotClipRunner
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
cpl = int(sys.argv[1]) if len(sys.argv) > 1 else 16
mode = sys.argv[2] if len(sys.argv) > 2 else "fp16x2"
r = SlotClipRunner(dev, 5, 1024, 2048, use_graph=False, clips_per_launch=cpl, input_form="nchw_f32" if mode in ("fp16x2", "fp32") else "tower16")
r.head.set_mode(mode)
r.load_clip(r.random_clip(1))
for _ in range(3):
    r.run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    r.run()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60, max_shapes_column_width=60))
# framework ops only (what is left of the framework inside a step), with the python line that issued them
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof2:
    r.run()
    torch.cuda.synchronize()
rows = [e for e in prof2.key_averages(group_by_stack_n=4) if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:30]:
    st = [s_ for s_ in e.stack if "slotvps_amd" in s_ or "bench" in s_][:2]
    print(f"{e.device_time_total / 1e3:8.3f} ms  x{e.count:3d}  {e.key:28s} {' <- '.join(x.strip()[-90:] for x in st)}")
