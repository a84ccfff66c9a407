#!/bin/bash
tools/ablate.sh 0:0 3:0 35:0 32:0
for d in 4 5; do
  touch slotvps_amd/csrc/retr_stats.hip
  make -C slotvps_amd/csrc EXTRA_retr_stats="-DSVPS_STATS_ABLATE -DSVPS_STA=$d" 2>&1 | grep -i "error"
  for a in 0 3; do echo "depth $d: $(SVPS_STATS_ABLATE=$a timeout -k 10 120 python tools/kbench_retr.py 2>&1 | tail -1)"; done
done
