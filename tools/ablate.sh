#!/bin/bash
# usage: tools/ablate.sh <stats ablation>:<retriever ablation> ...   - timing-only ablations of K3' / K1' (results wrong).
# Builds a SEPARATE library (libslotvps_hip_ablate.so, -DSVPS_STATS_ABLATE -DSVPS_RETR_ABLATE) and selects it through SLOTVPS_LIB:
# the product library is never touched and contains no ablation code.
cd "$(dirname "$0")/.."
make -C slotvps_amd/csrc ablate 2>&1 | grep -i "error"
for a in "$@"; do
  s=${a%%:*}; r=${a##*:}
  echo "$(SLOTVPS_LIB=$PWD/slotvps_amd/libslotvps_hip_ablate.so SVPS_STATS_ABLATE=$s SVPS_RETR_ABLATE=$r timeout -k 10 120 python tools/kbench_retr.py 2>&1 | tail -1)"
done
