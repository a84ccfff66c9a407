#!/bin/bash
# (GPU box) K4 timing-only ablations (results wrong) from a SEPARATE library (make ablk4 -> libslotvps_hip_ablk4.so, selected through
# SLOTVPS_LIB): the product library is never touched.
#   the eight-wave form v2 (SVPS_K4_V2=1 is set below):  1 no stores, 2 no MFMA, 4 no blend, 8 no staging
#   (the shipped wave-specialised v4 has s_memtime stamps instead: make stampk4, tools/k4_stamps.py)
# usage: bash tools/ablate_k4.sh 0 1 2 4 ...
make -C slotvps_amd/csrc ablk4 2>&1 | grep -i " error"
for a in "$@"; do
  echo "abl $a: $(SLOTVPS_LIB=slotvps_amd/libslotvps_hip_ablk4.so SVPS_K4_V2=1 SVPS_K4_ABLATE=$a timeout -k 10 200 python tools/kbench3.py --which k4 --levels 256x512 --T ${K4_T:-5} 2>&1 | grep nchw)"
done
