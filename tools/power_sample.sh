#!/bin/bash
# (GPU box) socket power and clocks while the bench step runs: python bench.py in the background, rocm-smi sampled every 0.5 s.
# usage: bash tools/power_sample.sh > gpurun_out/power.log
python bench.py --steps 900 --warmup 5 --cpu-baseline 0 --whole-detector 0 --latency-leg 0 --exact-leg 0 --viper-leg 0 > /tmp/power_bench.json 2> /tmp/power_bench.err &
BP=$!
sleep 28          # build of the runner, graph capture, warm-up
for i in $(seq 1 24); do
  if ! kill -0 $BP 2>/dev/null; then break; fi
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge)" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 0.5
done
wait $BP; echo "bench rc $?"; tail -5 /tmp/power_bench.err
echo "bench: $(tail -c 300 /tmp/power_bench.json | head -c 300)"
python -c "
import json; d=json.loads(open('/tmp/power_bench.json').read().strip().splitlines()[-1]); print('value', d['value'], 'ms_per_step', d['ms_per_step'])"
echo "idle:"; sleep 2; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr -s ' ' | tr '\n' ';'; echo
rocm-smi --showmaxpower 2>/dev/null | grep -i "power" | tr -s ' '
