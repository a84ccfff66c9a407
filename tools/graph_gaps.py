"""Idle time inside a hipGraph-replayed step: from a `rocprofv3 --kernel-trace --output-format csv` trace of
`bench.py --cpu-baseline 0 --whole-detector 0`, the gaps between consecutive kernels of the timed steps."""
import csv, glob, sys
fn = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(fn))]
rows.sort()
# steps are delimited by the mask decode kernel (last kernel of a step)
ends = [i for i, r in enumerate(rows) if "mask_decode" in r[2]]
steps = []
for a, b in zip(ends[:-1], ends[1:]):
    seg = rows[a + 1:b + 1]
    wall = seg[-1][1] - rows[a][1]
    busy = sum(e - s for s, e, _ in seg)
    gaps = sorted(((seg[i + 1][0] - seg[i][1]) / 1e3, seg[i][2][:40], seg[i + 1][2][:40]) for i in range(len(seg) - 1))
    steps.append((wall / 1e6, busy / 1e6, len(seg), gaps[-3:]))
mid = steps[len(steps) // 2:]           # the later (graph replay, timed) steps
for wall, busy, n, worst in mid[:6]:
    print(f"step wall {wall:.2f} ms, kernels busy {busy:.2f} ms ({100 * busy / wall:.1f} %), {n} kernels; largest gaps (us): "
          + "; ".join(f"{g:.1f} {a}->{b}" for g, a, b in worst))
