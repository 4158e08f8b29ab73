cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
python -m pytest tests/test_gpu_blackbox.py tests/test_gpu_edge_cases.py tests/test_gpu_switches.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -15 > gpurun_out/r02d/pytest.log
cat gpurun_out/r02d/pytest.log
python tools/closed_bench.py 2048 4096 8192 2>&1 | grep -v amdgpu > gpurun_out/r02d/closed.md
for o in 5 6; do MPK_LIB=$PWD/fancy_gym_amd/libmpk_occ$o.so python tools/closed_bench.py 2048 4096 8192 2>&1 | grep -v amdgpu | grep "auto\|lib" > gpurun_out/r02d/closed_occ$o.md; done
cat gpurun_out/r02d/closed.md gpurun_out/r02d/closed_occ5.md gpurun_out/r02d/closed_occ6.md
