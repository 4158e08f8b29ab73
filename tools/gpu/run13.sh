cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02m
export MPK_LIB=$PWD/fancy_gym_amd/libmpk_trace.so
python tools/dev/trace_kernel.py 4096 full 2>&1 | grep -v amdgpu | tail -60 > gpurun_out/r02m/trace_pipe_4096_full.txt
python tools/dev/trace_kernel.py 4096 plan 2>&1 | grep -v amdgpu | tail -60 > gpurun_out/r02m/trace_pipe_4096_plan.txt
python tools/dev/trace_kernel.py 8192 full pipe=1 2>&1 | grep -v amdgpu | tail -60 > gpurun_out/r02m/trace_pipe_8192_full.txt
unset MPK_LIB
python bench.py --no-cpu > gpurun_out/r02m/bench.json 2> gpurun_out/r02m/bench.err
timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r02m/fuzz.log
cat gpurun_out/r02m/trace_pipe_4096_full.txt gpurun_out/r02m/fuzz.log
