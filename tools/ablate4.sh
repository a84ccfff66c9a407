#!/bin/bash
touch slotvps_amd/csrc/retr_stats.hip
make -C slotvps_amd/csrc EXTRA_retr_stats="-DSVPS_STATS_ABLATE" 2>&1 | grep -i "error"
for T in 5 40; do
for a in 0:0 35:1; do
  s=${a%%:*}; r=${a##*:}
  echo "T=$T $(SVPS_STATS_ABLATE=$s SVPS_RETR_ABLATE=$r timeout -k 10 120 python tools/kbench_retr.py --T $T --iters 4 2>&1 | tail -1)"
done; done
