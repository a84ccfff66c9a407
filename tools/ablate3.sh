#!/bin/bash
for d in 0 1 0 1; do
  touch slotvps_amd/csrc/retr_stats.hip
  make -C slotvps_amd/csrc EXTRA_retr_stats="-DSVPS_STATS_ABLATE -DSVPS_STNT=$d" 2>&1 | grep -i "error"
  for a in 0 3; do echo "nt $d: $(SVPS_STATS_ABLATE=$a timeout -k 10 120 python tools/kbench_retr.py 2>&1 | tail -1)"; done
done
