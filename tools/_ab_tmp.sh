for v in sp4 sp3 sp2 a2 a2p a2p6 a2p2 sp4; do
  echo "== $v"; SLOTVPS_LIB=slotvps_amd/libslotvps_hip_v$v.so timeout -k 10 100 python tools/kbench_retr_hl.py --reps 3 2>&1 | grep rep | tail -2
done
