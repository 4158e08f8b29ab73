cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02i
python -m pytest tests -m gpu -q --deselect tests/test_gpu_fuzz.py 2>&1 | grep -E "passed|failed|Error|error|assert|FAILED" | tail -15 | tee gpurun_out/r02i/pytest.log
python tools/phase_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r02i/phase.md
