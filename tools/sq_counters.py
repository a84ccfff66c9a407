"""Summarise two rocprofv3 --pmc passes of SQ counters (tools/collect_profiles.sh, passes [7a] / [7b]) per library kernel:
matrix-pipe busy, vector-ALU busy, issue stalls, parked waves, LDS bank conflicts - the counters behind DESIGN.md's statements about
where the retriever / statistics kernels lose time.
    python3 tools/sq_counters.py <dir pass a> <dir pass b> <out.json>
Units (MI355X_MICROARCH.md, rocprofv3 PMC slots): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per v_mfma_f32_32x32x16); SQ_BUSY_CYCLES is per SE-summed busy time of the SQ."""
import csv
import glob
import json
import os
import sys

KERNELS = {"retr_attn_kernel": "retr_attn", "retr_attn_hl32_kernel": "retr_attn", "retr_stats_hl_kernel": "retr_stats", "retr_stats2_kernel": "retr_stats (level form, 2 stages)",
           "retr_stats_kernel": "retr_stats", "level_fuse": "level_fuse", "mask_decode": "mask_decode", "slot_ffn_kernel": "slot_ffn",
           "slot_chain_kernel": "slot_chain", "slot_gemm_kernel": "slot_gemm", "bgemm_kernel": "bgemm"}


def load(directory):
    out = {}
    for fn in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(fn) as fh:
            for row in csv.DictReader(fh):
                key = next((v for k, v in KERNELS.items() if k in row["Kernel_Name"]), None)
                if key is None:
                    continue
                rec = out.setdefault(key, {})
                c = rec.setdefault(row["Counter_Name"], [0.0, 0])
                c[0] += float(row["Counter_Value"])
                c[1] += 1
    return {k: {n: v[0] / v[1] for n, v in rec.items()} | {"launches": max(v[1] for v in rec.values())} for k, rec in out.items()}


def main():
    a, b, dst = sys.argv[1:4]
    ca, cb = load(a), load(b)
    res = {}
    for k in sorted(set(ca) | set(cb)):
        x = dict(ca.get(k, {}))
        x.update({n: v for n, v in cb.get(k, {}).items() if n != "launches"})
        d = {"launches_seen": x.get("launches"), "raw_per_launch": {n: round(v, 1) for n, v in x.items() if n != "launches"}}
        wc = x.get("SQ_WAVE_CYCLES")
        if wc:
            d["wave_cycles_share"] = {"issuing (ACTIVE_INST_ANY)": round(x.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3),
                                      "stalled at issue (WAIT_INST_ANY)": round(x.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                                      "parked (WAIT_ANY)": round(x.get("SQ_WAIT_ANY", 0) / wc, 3),
                                      "valu (ACTIVE_INST_VALU)": round(x.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3),
                                      "lds (ACTIVE_INST_LDS)": round(x.get("SQ_ACTIVE_INST_LDS", 0) / wc, 3)}
        if x.get("SQ_INSTS_MFMA") and x.get("SQ_INSTS_VALU"):
            d["valu_per_mfma"] = round(x["SQ_INSTS_VALU"] / x["SQ_INSTS_MFMA"], 2)
        if x.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_bank_conflict_share"] = round(x.get("SQ_LDS_BANK_CONFLICT", 0) / x["SQ_LDS_IDX_ACTIVE"], 4)
        if x.get("SQ_VALU_MFMA_BUSY_CYCLES") and x.get("SQ_VALU_MFMA_COEXEC_CYCLES") is not None:
            d["mfma_busy_cycles_with_valu_coexec_share"] = round(x["SQ_VALU_MFMA_COEXEC_CYCLES"] / x["SQ_VALU_MFMA_BUSY_CYCLES"], 3)
        res[k] = d
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    with open(dst, "w") as fh:
        cmd = sys.argv[4] if len(sys.argv) > 4 else ("rocprofv3 --kernel-trace --pmc <8 SQ counters per pass, two passes> -- python3 bench.py --steps 2 --warmup 1 "
                                                     "--cpu-baseline 0 --whole-detector 0 --latency-leg 0 --exact-leg 0 --viper-leg 0 --no-graph "
                                                     "(tools/collect_profiles.sh, passes [7a] / [7b])")
        json.dump({"command": cmd, "kernels": res}, fh, indent=1)
    print(json.dumps({k: v.get("wave_cycles_share") for k, v in res.items()}))


if __name__ == "__main__":
    main()
