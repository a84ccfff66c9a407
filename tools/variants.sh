#!/bin/bash
# usage: tools/variants.sh <file-stem> <macro> v1 v2 ...   - rebuilds one source with -D<macro>=v and times K3'/K1' (GPU box)
stem=$1; macro=$2; shift 2
for v in "$@"; do
  touch slotvps_amd/csrc/$stem.hip
  make -C slotvps_amd/csrc EXTRA_$stem="-D$macro=$v" 2>&1 | grep -i "error"
  echo "$macro=$v: $(timeout -k 10 120 python tools/kbench_retr.py 2>&1 | tail -1)"
done
