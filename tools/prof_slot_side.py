"""torch.profiler view of one eager clip step: which host ops launch the small slot-side kernels."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd.clip import SlotClipRunner
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
cpl = int(sys.argv[1]) if len(sys.argv) > 1 else 16
r = SlotClipRunner(dev, 5, 1024, 2048, use_graph=False, clips_per_launch=cpl)
r.load_clip(r.random_clip(1))
for _ in range(3):
    r.run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    r.run()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60, max_shapes_column_width=60))
