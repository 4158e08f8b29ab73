cd $GRAFT_REPO_ROOT
O=gpurun_out/r02v; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -12 > $O/pytest.log
python tools/phase_bench.py 4096 65536 2>&1 | grep -v amdgpu | grep "cfg" > $O/phase.md
MPK_LIB=$PWD/fancy_gym_amd/libmpk_trace.so python tools/dev/trace_phase.py 65536 2>&1 | grep -v amdgpu | tail -14 > $O/trace_phase.txt
cat $O/pytest.log $O/phase.md $O/trace_phase.txt
