#!/bin/bash
# A / B libraries for same-box kernel comparisons: tools/ab_build.sh <source.hip> NAME:"-Dflags" [NAME:"-Dflags" ...]
# builds slotvps_amd/libslotvps_hip_v<NAME>.so = the product library with <source.hip> compiled with the extra flags (select with SLOTVPS_LIB).
set -e
cd "$(dirname "$0")/../slotvps_amd/csrc"
src=$1; shift
make -s all
CX="-O3 -std=c++17 -fPIC -fno-slp-vectorize --offload-arch=gfx950"
SRCS=$(sed -n 's/^SRCS := \(abi.hip.*\)$/\1/p' Makefile)
OBJS=""
for f in $SRCS; do [ "$f" != "$src" ] && OBJS="$OBJS build/${f%.hip}.o"; done
for v in "$@"; do
  n=${v%%:*}; fl=${v#*:}
  /opt/rocm/bin/hipcc $CX $fl -c $src -o build/${src%.hip}_v$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS build/${src%.hip}_v$n.o -o ../libslotvps_hip_v$n.so
  echo "built libslotvps_hip_v$n.so ($fl)"
done
