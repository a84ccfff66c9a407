"""Summarise the HBM-traffic PMC passes of bench.py into profiles/<round>/pmc_traffic.json, per kernel, CALIBRATED in the
same pass on a kernel whose bytes are known.

Collect on the GPU box (counters in their own runs, kernel-trace only - MI355X_MICROARCH.md, HBM / rocprofv3 sections; the
program itself after `--`, no launcher in between):
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch -o f --output-format csv -- \
        python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-baseline 0 --whole-detector 0 --latency-leg 0 --no-graph
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write -o w --output-format csv -- (same command)
    python3 tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r05/pmc_traffic.json "<workload key>" <mode>

Units and corrections. The counters come in KiB (value * 1024 B). The microarchitecture guide states that on gfx950
FETCH_SIZE tallies the 128-byte requests of wide coalesced streams at 64 B (so it reads half the bytes) and that WRITE_SIZE is
exact. Instead of trusting that factor blindly it is MEASURED here: bench.py's copy-ceiling leg launches the library's own
known-bytes streaming kernel (svps_probe_copy: 16 B per lane loads and stores, the access shape of the library's streams,
13 launches reading 2^30 and writing 2^30 bytes each) inside the same profiled process; `calibration.fetch_factor` = known bytes / reported bytes of those launches (expected: 2.0), likewise for writes
(expected: 1.0). Both the raw and the calibrated numbers are stored."""
import csv
import glob
import hashlib
import json
import os
import sys

# kernel-name fragment in the trace -> the name bench.py uses in `roofline.per_kernel`
KERNELS = {"retr_attn_kernel": "retr_attn", "retr_attn_hl32_kernel": "retr_attn", "retr_stats_kernel": "retr_stats", "retr_stats_hl_kernel": "retr_stats", "retr_stats2_kernel": "retr_stats", "retr_logit_stats": "retr_logit_stats", "retr_probs_kernel": "retr_probs", "retr_pv_kernel": "retr_pv",
           "retr_finish": "retr_finish", "slot_attn_partial": "slot_attn", "slot_attn_finish": "slot_attn_finish",
           "kv_project": "kv_project", "level_fuse": "level_fuse", "mask_decode": "mask_decode", "row_ln": "row_ln",
           "slot_self_attn": "slot_self_attn"}
COPY_BYTES = 1 << 30


def per_kernel(directory, counter):
    out, copies = {}, []
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {directory}")
    for fn in files:
        with open(fn) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name, val = row["Kernel_Name"], float(row["Counter_Value"])
                key = next((v for k, v in KERNELS.items() if k in name), None)
                if key is not None:
                    rec = out.setdefault(key, {"launches": 0, "sum_kb": 0.0})
                    rec["launches"] += 1
                    rec["sum_kb"] += val
                elif "probe_copy_kernel" in name:
                    copies.append(val * 1024)            # the library's known-bytes streaming kernel: 1 GiB in, 1 GiB out per launch
    return out, copies


def main():
    fetch_dir, write_dir, dst = sys.argv[1:4]
    fetch, fcop = per_kernel(fetch_dir, "FETCH_SIZE")
    write, wcop = per_kernel(write_dir, "WRITE_SIZE")
    if not fcop or not wcop:
        raise SystemExit("calibration launches (1 GiB torch copy) not found in the counter CSVs")
    ff = COPY_BYTES / (sum(fcop) / len(fcop))
    wf = COPY_BYTES / (sum(wcop) / len(wcop))
    kernels = {}
    for k in fetch:
        n = fetch[k]["launches"]
        raw_f = fetch[k]["sum_kb"] * 1024 / n
        raw_w = write.get(k, {"sum_kb": 0.0, "launches": n})["sum_kb"] * 1024 / max(1, write.get(k, {"launches": n})["launches"])
        kernels[k] = {"launches": n, "fetch_bytes_per_launch_raw": raw_f, "write_bytes_per_launch_raw": raw_w,
                      "fetch_bytes_per_launch": raw_f * ff, "write_bytes_per_launch": raw_w * wf,
                      "traffic_bytes_per_launch": raw_f * ff + raw_w * wf}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "bench.py"), "rb") as fh:
        sha = hashlib.sha256(fh.read()).hexdigest()[:16]
    rec = {"calibration_kernel": "svps::probe_copy_kernel (csrc/diag_probes.hip)", "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 --warmup 1 "
                      "--cpu-baseline 0 --whole-detector 0 --latency-leg 0 --no-graph",
           "bench_sha": sha,
           "calibration": {"what": "1 GiB copies by svps::probe_copy_kernel launched by the same process (bench.py copy-ceiling leg)",
                           "copy_launches_seen": [len(fcop), len(wcop)], "fetch_factor": ff, "write_factor": wf,
                           "expected": "fetch 2.0 (gfx950 tallies 128-B requests at 64 B), write 1.0"},
           "kernels": kernels}
    if len(sys.argv) > 4:
        rec["workload_key"] = sys.argv[4]
    rec["mode"] = sys.argv[5] if len(sys.argv) > 5 else "bf16"          # head mode of the profiled step (bench.py --mode)
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    with open(dst, "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps({"fetch_factor": ff, "write_factor": wf,
                      **{k: round(v["traffic_bytes_per_launch"] / 1e9, 3) for k, v in kernels.items()}}))


if __name__ == "__main__":
    main()
