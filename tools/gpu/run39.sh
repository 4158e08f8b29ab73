cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_blackbox.py -m gpu -x -q -k "rccl or bench_gpus" 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | tail -5
