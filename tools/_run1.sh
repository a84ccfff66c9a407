cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 300 python -m pytest tests/test_detector.py -x -q -m gpu -k "graph_follows" > gpurun_out/r3/t12.log 2>&1; tail -3 gpurun_out/r3/t12.log
timeout -k 10 900 python bench.py > gpurun_out/r3/bench_a.json 2> gpurun_out/r3/bench_a.err; echo "bench rc=$?"; tail -5 gpurun_out/r3/bench_a.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3/bench_a.json'))
print(d['value'], d['ms_per_step'])
r=d['roofline']
for k,v in r['per_kernel'].items(): print(k, v.get('launches'), v.get('avg_launch_us'), v.get('hbm_frac'), v.get('mfma_frac'), v.get('mfma_frac_executed'))
print(r.get('retriever_pair'))
for k in ('cpu_baseline','single_clip_latency_ms','exact_mode','other_configs','whole_detector'): print(k, d.get(k))
PY
