"""(GPU box) The reference-precision leg of bench.py on its own: frames/s of the fp16x2 step + its per-kernel roofline object.
    python tools/refprec_bench.py [--clips-per-launch 32] [--steps 3] [--precision fp16x2]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--clips-per-launch", type=int, default=32)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--precision", default="fp16x2")
ap.add_argument("--height", type=int, default=1024)
ap.add_argument("--width", type=int, default=2048)
ap.add_argument("--frames", type=int, default=5)
ap.add_argument("--slots", type=int, default=100)
x = ap.parse_args()
sys.argv = [sys.argv[0]]
a = bench.parse()
a.height, a.width, a.frames, a.slots = x.height, x.width, x.frames, x.slots
dev = torch.device("cuda:0")
res = bench.side_leg(a, dev, a.frames, a.height, a.width, a.slots, a.num_classes, x.clips_per_launch, x.steps, mode=x.precision, with_roofline=True)
print(json.dumps(res), flush=True)
