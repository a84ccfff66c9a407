cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3/gpu_all_1.log 2>&1; tail -4 gpurun_out/r3/gpu_all_1.log
