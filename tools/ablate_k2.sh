#!/bin/bash
# (GPU box) K2 argmax-only mode, timing-only ablations (results wrong) from a SEPARATE library (make ablk2 -> libslotvps_hip_ablk2.so,
# selected through SLOTVPS_LIB): the product library is never touched.
#   1 no MFMAs, 2 no fragment reads either, 4 no argmax epilogue, 8 no DMA (combinations 5, 6, 14); 16 three-tile ring; 32 the un-skewed loop
# usage: bash tools/ablate_k2.sh 0 1 2 4 ...
make -C slotvps_amd/csrc ablk2 2>&1 | grep -i " error"
for a in "$@"; do
  echo "abl $a: $(SLOTVPS_LIB=slotvps_amd/libslotvps_hip_ablk2.so SVPS_K2_ABLATE=$a timeout -k 10 200 python tools/kbench3.py --which k2 --levels 256x512 --T ${K2_T:-40} 2>&1 | grep 'argmax only')"
done
