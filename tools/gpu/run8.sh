cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02h
python -m pytest tests/test_gpu_blackbox.py tests/test_gpu_switches.py -m gpu -q 2>&1 | grep -E "passed|failed|Error|error|assert|FAILED" | tail -15 | tee gpurun_out/r02h/pytest.log
python tools/closed_bench.py 2048 4096 8192 2>&1 | grep -v amdgpu | grep "auto\|lib\|pipe=0\|split" | tee gpurun_out/r02h/closed.md
for B in 2048 8192; do python tools/bench_replan.py $B 50 --graph; done 2>&1 | grep -v amdgpu | tee gpurun_out/r02h/replan.log
