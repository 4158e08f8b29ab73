cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
python -m pytest tests/test_gpu_blackbox.py tests/test_gpu_edge_cases.py tests/test_gpu_switches.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -15 > gpurun_out/r02c/pytest.log
cat gpurun_out/r02c/pytest.log
MPK_LIB=$PWD/fancy_gym_amd/libmpk_trace.so python tools/dev/trace_kernel.py 4096 full quad=3 2>&1 | grep -v amdgpu | tail -27
python tools/closed_bench.py 4096 8192 2>&1 | grep -v amdgpu > gpurun_out/r02c/closed.md
MPK_LIB=$PWD/fancy_gym_amd/libmpk_occ4.so python tools/closed_bench.py 4096 8192 2>&1 | grep -v amdgpu | grep "auto\|lib" > gpurun_out/r02c/closed_occ4.md
cat gpurun_out/r02c/closed.md gpurun_out/r02c/closed_occ4.md
