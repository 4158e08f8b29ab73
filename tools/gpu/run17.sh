cd $GRAFT_REPO_ROOT
O=gpurun_out/r02q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_blackbox.py tests/test_gpu_fuzz.py tests/test_gpu_switches.py tests/test_gpu_ode.py -m gpu -x -q 2>&1 | tail -15 > $O/pytest.log
MPK_LIB=$PWD/fancy_gym_amd/libmpk_trace.so python tools/dev/trace_kernel.py 4096 full 2>&1 | grep -v amdgpu | tail -60 > $O/trace_base.txt
MPK_LIB=$PWD/fancy_gym_amd/libmpk_trace.so python tools/dev/trace_kernel.py 4096 plan 2>&1 | grep -v amdgpu | tail -60 > $O/trace_plan.txt
python tools/closed_bench.py 2048 4096 8192 16384 2>&1 | grep -v amdgpu | grep "lib\|auto\|pipe=1\|duo" > $O/closed.md
cat $O/pytest.log $O/closed.md; paste $O/trace_base.txt $O/trace_plan.txt | cut -c1-200
