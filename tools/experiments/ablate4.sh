#!/bin/bash
# Timing-only ablations of the four-wave retriever (csrc/retr_attn4.hip, R4_ABL bits): one process per library variant.
#   make -C slotvps_amd/csrc abl4 && bash tools/ablate4.sh [kbench_retr arguments]
cd "$(dirname "$0")/.."
echo "== full kernel"; python tools/kbench_retr.py --form w4 --reps 2 "$@" | grep rep
for n in ${ABL4:-1 2 4 8 16 32 64 63}; do
  lib=slotvps_amd/libslotvps_hip_abl$n.so
  [ -f $lib ] || continue
  echo "== R4_ABL=$n"; SLOTVPS_LIB=$PWD/$lib python tools/kbench_retr.py --form w4 --reps 2 "$@" | grep rep
done
