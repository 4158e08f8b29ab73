cd $GRAFT_REPO_ROOT
O=gpurun_out/r02p; mkdir -p $O
MPK_LIB=$PWD/fancy_gym_amd/libmpk_trace.so python tools/dev/trace_kernel.py 4096 full 2>&1 | grep -v amdgpu | tail -60 > $O/trace_base.txt
MPK_LIB=$PWD/fancy_gym_amd/libmpk_starve_trace.so python tools/dev/trace_kernel.py 4096 full 2>&1 | grep -v amdgpu | tail -60 > $O/trace_starve.txt
MPK_LIB=$PWD/fancy_gym_amd/libmpk_starve_trace.so python tools/dev/trace_kernel.py 256 full pipe=1 2>&1 | grep -v amdgpu | tail -60 > $O/trace_starve_256.txt
MPK_LIB=$PWD/fancy_gym_amd/libmpk_trace.so python tools/dev/trace_kernel.py 256 full pipe=1 2>&1 | grep -v amdgpu | tail -60 > $O/trace_base_256.txt
timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -30 > $O/fuzz.log
paste $O/trace_base.txt $O/trace_starve.txt | cut -c1-200; paste $O/trace_base_256.txt $O/trace_starve_256.txt | cut -c1-200; tail -5 $O/fuzz.log
