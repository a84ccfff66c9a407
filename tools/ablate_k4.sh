#!/bin/bash
# (GPU box) K4 timing-only ablations: library built with -DSVPS_K4_ABLATE, SVPS_K4_ABLATE = 1 no stores, 2 no MFMA, 4 no blend, 8 no staging
touch slotvps_amd/csrc/level_fuse.hip
make -C slotvps_amd/csrc EXTRA_level_fuse="-DSVPS_K4_ABLATE" 2>&1 | grep -i " error"
for a in "$@"; do
  echo "abl $a: $(SVPS_K4_ABLATE=$a timeout -k 10 200 python tools/kbench3.py --which k4 --levels 256x512 2>&1 | grep nchw)"
done
