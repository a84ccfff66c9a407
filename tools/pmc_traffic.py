"""Summarise the HBM-traffic PMC passes of bench.py into profiles/<round>/k1_pmc_traffic.json.

Collect (GPU box; counters in their own runs, kernel-trace only - MI355X_MICROARCH.md, HBM section):
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch -o f --output-format csv -- \
        python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-baseline 0 --no-graph
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write -o w --output-format csv -- \
        python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-baseline 0 --no-graph
    python3 tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01/k1_pmc_traffic.json
FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1 KB = 1024 B per the counter definition (value * 1024 B);
gfx950 correction: FETCH_SIZE tallies the 128-B requests of wide coalesced streams at 64 B -> doubled."""
import csv
import glob
import json
import os
import sys


def per_kernel(directory, counter):
    out = {}
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {directory}")
    for fn in files:
        with open(fn) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = row["Kernel_Name"]
                key = next((k for k in ("slot_attn_partial", "slot_attn_finish", "kv_project", "level_fuse", "mask_decode",
                                        "row_ln") if k in name), None)
                if key is None:
                    continue
                rec = out.setdefault(key, {"launches": 0, "sum_kb": 0.0})
                rec["launches"] += 1
                rec["sum_kb"] += float(row["Counter_Value"])
    return out


def main():
    fetch_dir, write_dir, dst = sys.argv[1:4]
    fetch = per_kernel(fetch_dir, "FETCH_SIZE")
    write = per_kernel(write_dir, "WRITE_SIZE")
    k = "slot_attn_partial"
    n = fetch[k]["launches"]
    assert n == write[k]["launches"], (n, write[k]["launches"])
    fb = fetch[k]["sum_kb"] * 1024 * 2 / n
    wb = write[k]["sum_kb"] * 1024 / n
    rec = {"kernel": "slot_attn_partial_ws",
           "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 "
                      "--warmup 1 --cpu-baseline 0 --no-graph (default workload: 16 clips of T=5 stacked per launch)",
           "launches": n, "fetch_bytes_per_launch_corrected_x2": fb, "write_bytes_per_launch": wb,
           "traffic_bytes_per_launch": fb + wb,
           "note": "gfx950: FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced streams -> doubled "
                   "(MI355X_MICROARCH.md HBM section); WRITE_SIZE exact",
           "per_kernel_traffic_bytes_per_launch": {
               kk: (fetch[kk]["sum_kb"] * 2048 + write.get(kk, {"sum_kb": 0.0})["sum_kb"] * 1024) / fetch[kk]["launches"]
               for kk in fetch},
           "all_kernels": {"FETCH_SIZE": fetch, "WRITE_SIZE": write}}
    if len(sys.argv) > 4:
        rec["workload_key"] = sys.argv[4]
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    with open(dst, "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps({kk: rec[kk] for kk in ("launches", "fetch_bytes_per_launch_corrected_x2", "write_bytes_per_launch",
                                             "traffic_bytes_per_launch")}))


if __name__ == "__main__":
    main()
