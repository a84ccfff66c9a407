#!/bin/bash
# (GPU box) K4 timing-only ablations (results wrong): the library is rebuilt with -DSVPS_K4_ABLATE on the box only.
#   the eight-wave form v2 (SVPS_K4_V2=1 is set below):  1 no stores, 2 no MFMA, 4 no blend, 8 no staging
#   (the shipped wave-specialised v4 has s_memtime stamps instead: make stampk4, tools/k4_stamps.py)
# usage: bash tools/ablate_k4.sh 0 1 2 4 ...
touch slotvps_amd/csrc/level_fuse.hip
make -C slotvps_amd/csrc EXTRA_level_fuse="-DSVPS_K4_ABLATE" 2>&1 | grep -i " error"
for a in "$@"; do
  echo "abl $a: $(SVPS_K4_V2=1 SVPS_K4_ABLATE=$a timeout -k 10 200 python tools/kbench3.py --which k4 --levels 256x512 --T ${K4_T:-5} 2>&1 | grep nchw)"
done
