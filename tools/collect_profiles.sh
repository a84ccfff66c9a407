# Collects everything profiles/<round>/ holds from ONE box: bench line, rocprofv3 kernel stats, both PMC passes, VIPER line, whole-detector lines.
# usage (GPU box): bash tools/collect_profiles.sh   -> gpurun_out/v7/
set -o pipefail
R=$GRAFT_REPO_ROOT
V=v7
O=$R/gpurun_out/$V
mkdir -p $O
cd $R
echo "[1] bench"; timeout -k 10 420 python3 bench.py > $O/bench_$V.json 2> $O/bench_$V.err || echo "bench rc $?"
echo "[2] kernel stats"; cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/stats -o b --output-format csv -- python3 $R/bench.py --cpu-baseline 0 --whole-detector 0 --latency-leg 0 --no-graph --steps 5 --warmup 2 > $O/bench_under_rocprof_$V.json 2> $O/stats.err || echo "stats rc $?"
echo "[3] pmc fetch"; timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-baseline 0 --whole-detector 0 --latency-leg 0 --no-graph > $O/pmc_fetch.json 2> $O/pmc_fetch.err || echo "fetch rc $?"
echo "[4] pmc write"; timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-baseline 0 --whole-detector 0 --latency-leg 0 --no-graph > $O/pmc_write.json 2> $O/pmc_write.err || echo "write rc $?"
cd $R
echo "[5] viper"; timeout -k 10 200 python3 bench.py --height 1088 --width 1920 --frames 10 --slots 200 --num-classes 24 --clips-per-launch 8 --steps 6 --warmup 2 --cpu-baseline 0 --whole-detector 0 --latency-leg 0 > $O/bench_viper_$V.json 2> $O/viper.err || echo "viper rc $?"
echo "[6] e2e"; for c in r50_fpn_slotvps_mi355x swinL_fpn_slotvps_mi355x viper_r50_slotvps_mi355x; do timeout -k 10 200 python3 tools/detector_e2e.py --config configs/$c.py >> $O/whole_detector_configs.jsonl 2>> $O/e2e.err; echo "e2e $c done"; done
ls -la $O
