"""Builds the bench's clip runner and captures + validates its hipGraph (SlotClipRunner.run), once; exit code 1 with the mismatch report
if the first replay differs from the eager step. Meant to be run many times in a row (fresh process each time)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd import synth
from slotvps_amd.clip import SlotClipRunner
dev = torch.device("cuda:0")
cpl = int(os.environ.get("CPL", "32"))
r = SlotClipRunner(dev, 5, 1024, 2048, L=100, param_seed=0, cfg=dict(synth.R50_HEAD_CFG, num_classes=20), use_graph=True, n_slots=2, clips_per_launch=cpl)
for i in range(2):
    r.load_clip(r.random_clip(1234 + i), slot=i)
try:
    for i in range(4):
        r.run(slot=i % 2)
    torch.cuda.synchronize()
    print("ok")
except RuntimeError as e:
    print("MISMATCH", str(e)[:3000])
    sys.exit(1)
