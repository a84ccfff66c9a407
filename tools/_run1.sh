cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 400 python -m pytest tests/test_head_gpu.py -x -q -m gpu -s > gpurun_out/r3/t10.log 2>&1; grep -E "free-running|slot argmax equal|passed|failed|Error" gpurun_out/r3/t10.log | head -20
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
