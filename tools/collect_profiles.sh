# Collects everything profiles/<round>/ holds from ONE box: bench line, rocprofv3 kernel stats, both HBM-traffic PMC passes, two SQ
# counter passes on the retriever / statistics kernels (all on the headline mode fp16x2), kernel stats of mode bf16, VIPER line, whole-detector lines.
# usage (GPU box): [PARTS="1 2 3 4"] bash tools/collect_profiles.sh [tag]   -> gpurun_out/<tag>/     (PARTS: which steps; default all)
set -o pipefail
R=$GRAFT_REPO_ROOT
V=${1:-r05a}
PARTS=${PARTS:-"1 2 3 4 7 8 5 6 9"}
O=$R/gpurun_out/$V
mkdir -p $O
cd $R
LEGS="--cpu-baseline 0 --whole-detector 0 --latency-leg 0 --exact-leg 0 --viper-leg 0 --parity-leg 0"
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
has 1 && { echo "[1] bench"; timeout -k 10 700 python3 bench.py > $O/bench_$V.json 2> $O/bench_$V.err || echo "bench rc $?"; }
cd /tmp && export TMPDIR=/tmp
has 2 && { echo "[2] kernel stats"; timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/stats -o b --output-format csv -- python3 $R/bench.py $LEGS --no-graph --steps 5 --warmup 2 > $O/bench_under_rocprof_$V.json 2> $O/stats.err || echo "stats rc $?"; }
has 3 && { echo "[3] pmc fetch"; timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 $LEGS --no-graph > $O/pmc_fetch.json 2> $O/pmc_fetch.err || echo "fetch rc $?"; }
has 4 && { echo "[4] pmc write"; timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 $LEGS --no-graph > $O/pmc_write.json 2> $O/pmc_write.err || echo "write rc $?"; }
has 7 && { echo "[7a] SQ counters a"; timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d $O/sq_a -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 $LEGS --no-graph > $O/sq_a.json 2> $O/sq_a.err || echo "sq a rc $?"
          echo "[7b] SQ counters b"; timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM -d $O/sq_b -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 $LEGS --no-graph > $O/sq_b.json 2> $O/sq_b.err || echo "sq b rc $?"; }
has 8 && { echo "[8] mode bf16 (round 4's headline definition): kernel stats"; timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/stats_bf16 -o b --output-format csv -- python3 $R/bench.py --mode bf16 $LEGS --no-graph --steps 5 --warmup 2 > $O/bench_bf16_under_rocprof_$V.json 2> $O/stats_bf16.err || echo "bf16 stats rc $?"; }
cd $R
has 5 && { echo "[5] viper"; timeout -k 10 200 python3 bench.py --height 1088 --width 1920 --frames 10 --slots 200 --num-classes 24 --clips-per-launch 8 --steps 6 --warmup 2 $LEGS > $O/bench_viper_$V.json 2> $O/viper.err || echo "viper rc $?"; }
has 9 && { echo "[9] bench --legs all"; timeout -k 10 900 python3 bench.py --legs all > $O/bench_all_legs_$V.json 2> $O/bench_all_legs_$V.err || echo "bench all rc $?"; }
has 6 && { echo "[6] e2e (the configs' own head mode fp16x2, then bf16)"; for c in r50_fpn_slotvps_mi355x; do timeout -k 10 200 python3 tools/detector_e2e.py --config configs/$c.py --mode bf16 >> $O/whole_detector_configs_mode_bf16.jsonl 2>> $O/e2e.err; done; for c in r50_fpn_slotvps_mi355x swinL_fpn_slotvps_mi355x viper_r50_slotvps_mi355x; do timeout -k 10 200 python3 tools/detector_e2e.py --config configs/$c.py >> $O/whole_detector_configs.jsonl 2>> $O/e2e.err; echo "e2e $c done"; done; }
has 3 && has 4 && { python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json "1024x2048 T=5 L=100 cpl=32" fp16x2 > $O/pmc_traffic.log 2>&1; cat $O/pmc_traffic.log; }
has 7 && { python3 tools/sq_counters.py $O/sq_a $O/sq_b $O/sq_counters.json > $O/sq_counters.log 2>&1; cat $O/sq_counters.log; }
# keep the merge-back small: the raw counter CSVs are large
find $O -name "*counter_collection.csv" -size +20M -delete; find $O -name "*kernel_trace.csv" -size +20M -delete
ls -la $O
