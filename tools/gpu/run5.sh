cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02e
python -m pytest tests/test_gpu_blackbox.py tests/test_gpu_edge_cases.py tests/test_gpu_switches.py -m gpu -q 2>&1 | grep -E "passed|failed|Error|error|assert|FAILED" | tail -25 > gpurun_out/r02e/pytest.log
cat gpurun_out/r02e/pytest.log
python tools/closed_bench.py 2048 4096 8192 2>&1 | grep -v amdgpu > gpurun_out/r02e/closed.md
cat gpurun_out/r02e/closed.md
for B in 2048 8192; do python tools/bench_replan.py $B 50 --graph; done 2>&1 | grep -v amdgpu > gpurun_out/r02e/replan.log
cat gpurun_out/r02e/replan.log
