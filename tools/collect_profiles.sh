# Collects everything profiles/<round>/ holds from ONE box: bench line, rocprofv3 kernel stats, both HBM-traffic PMC passes, two SQ
# counter passes on the retriever / statistics kernels, VIPER line, whole-detector lines.
# usage (GPU box): bash tools/collect_profiles.sh [tag]   -> gpurun_out/<tag>/
set -o pipefail
R=$GRAFT_REPO_ROOT
V=${1:-r03a}
O=$R/gpurun_out/$V
mkdir -p $O
cd $R
LEGS="--cpu-baseline 0 --whole-detector 0 --latency-leg 0 --exact-leg 0 --viper-leg 0"
echo "[1] bench"; timeout -k 10 600 python3 bench.py > $O/bench_$V.json 2> $O/bench_$V.err || echo "bench rc $?"
echo "[2] kernel stats"; cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/stats -o b --output-format csv -- python3 $R/bench.py $LEGS --no-graph --steps 5 --warmup 2 > $O/bench_under_rocprof_$V.json 2> $O/stats.err || echo "stats rc $?"
echo "[3] pmc fetch"; timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 $LEGS --no-graph > $O/pmc_fetch.json 2> $O/pmc_fetch.err || echo "fetch rc $?"
echo "[4] pmc write"; timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 $LEGS --no-graph > $O/pmc_write.json 2> $O/pmc_write.err || echo "write rc $?"
echo "[7a] SQ counters a"; timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d $O/sq_a -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 $LEGS --no-graph > $O/sq_a.json 2> $O/sq_a.err || echo "sq a rc $?"
echo "[7b] SQ counters b"; timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM -d $O/sq_b -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 $LEGS --no-graph > $O/sq_b.json 2> $O/sq_b.err || echo "sq b rc $?"
cd $R
echo "[5] viper"; timeout -k 10 200 python3 bench.py --height 1088 --width 1920 --frames 10 --slots 200 --num-classes 24 --clips-per-launch 8 --steps 6 --warmup 2 $LEGS > $O/bench_viper_$V.json 2> $O/viper.err || echo "viper rc $?"
echo "[6] e2e"; for c in r50_fpn_slotvps_mi355x swinL_fpn_slotvps_mi355x viper_r50_slotvps_mi355x; do timeout -k 10 200 python3 tools/detector_e2e.py --config configs/$c.py >> $O/whole_detector_configs.jsonl 2>> $O/e2e.err; echo "e2e $c done"; done
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json "1024x2048 T=5 L=100 cpl=32" > $O/pmc_traffic.log 2>&1; cat $O/pmc_traffic.log
python3 tools/sq_counters.py $O/sq_a $O/sq_b $O/sq_counters.json > $O/sq_counters.log 2>&1; cat $O/sq_counters.log
# keep the merge-back small: the raw counter CSVs are large
find $O -name "*counter_collection.csv" -size +20M -delete; find $O -name "*kernel_trace.csv" -size +20M -delete
ls -la $O
