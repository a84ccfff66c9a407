cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 200 python -m pytest tests/test_retr_fused_gpu.py -x -q -m gpu > gpurun_out/r3/t7.log 2>&1; tail -2 gpurun_out/r3/t7.log
timeout -k 10 100 python tools/kbench_retr.py --form w4 --reps 3 2>&1 | grep rep
timeout -k 10 100 python tools/kbench_retr.py --form w8 --reps 2 2>&1 | grep rep
