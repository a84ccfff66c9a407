#!/bin/bash
# usage: tools/ablate.sh <stats ablations...>   - library built with -DSVPS_STATS_ABLATE; K1' ablations via second arg "s:a"
touch slotvps_amd/csrc/retr_stats.hip
make -C slotvps_amd/csrc EXTRA_retr_stats="-DSVPS_STATS_ABLATE" 2>&1 | grep -i "error"
for a in "$@"; do
  s=${a%%:*}; r=${a##*:}
  echo "$(SVPS_STATS_ABLATE=$s SVPS_RETR_ABLATE=$r timeout -k 10 120 python tools/kbench_retr.py 2>&1 | tail -1)"
done
