cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_blackbox.py -m gpu -x -q -k "bench_line" 2>&1 | grep -E "passed|failed|FAILED|Error|assert|KeyError" | tail -8
