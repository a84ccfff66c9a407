cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
for n in k1_95 k0_95 k1_0 k0_0; do
  echo "== $n"; SLOTVPS_LIB=$PWD/slotvps_amd/libslotvps_hip_abl$n.so timeout -k 10 100 python tools/kbench_retr.py --form w4 --reps 2 2>&1 | grep "rep 1" | sed -e 's/retr_stats.*retr_attn/retr_attn/'
done > gpurun_out/r3/abl4_f.log 2>&1
cat gpurun_out/r3/abl4_f.log
