cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 400 python -m pytest tests/test_retr_robust_gpu.py -q -m gpu -s > gpurun_out/r3/t11.log 2>&1; grep -E "err|passed|failed|Error|assert" gpurun_out/r3/t11.log | head -40
