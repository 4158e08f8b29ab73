cd $GRAFT_REPO_ROOT
O=gpurun_out/r02s; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_blackbox.py tests/test_gpu_fuzz.py tests/test_gpu_switches.py tests/test_gpu_ode.py tests/test_gpu_edge_cases.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -8 > $O/pytest.log
cat $O/pytest.log
