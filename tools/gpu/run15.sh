cd $GRAFT_REPO_ROOT
O=gpurun_out/r02o; mkdir -p $O
MPK_LIB=$PWD/fancy_gym_amd/libmpk_trace.so python tools/dev/trace_kernel.py 4096 full 2>&1 | grep -v amdgpu | tail -50 > $O/trace_base.txt
MPK_LIB=$PWD/fancy_gym_amd/libmpk_idle_trace.so python tools/dev/trace_kernel.py 4096 full 2>&1 | grep -v amdgpu | tail -50 > $O/trace_idle.txt
python tools/closed_bench.py 2048 4096 8192 2>&1 | grep -v amdgpu | grep "lib\|auto\|pipe=1" > $O/closed_base.md
MPK_LIB=$PWD/fancy_gym_amd/libmpk_idle.so python tools/closed_bench.py 2048 4096 8192 2>&1 | grep -v amdgpu | grep "lib\|auto\|pipe=1" > $O/closed_idle.md
timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -30 > $O/fuzz.log
paste $O/trace_base.txt $O/trace_idle.txt | cut -c1-200; cat $O/closed_base.md $O/closed_idle.md $O/fuzz.log
