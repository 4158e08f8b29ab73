cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02g
python tools/occ_probe.py 2>&1 | grep -v amdgpu | tee gpurun_out/r02g/occ_probe.md
MPK_LIB=$PWD/fancy_gym_amd/libmpk_trace.so python tools/dev/trace_kernel.py 4096 full 2>&1 | grep -v amdgpu | tail -19
python tools/closed_bench.py 2048 4096 8192 2>&1 | grep -v amdgpu | grep "auto\|lib\|split=0" | tee gpurun_out/r02g/closed.md
MPK_LIB=$PWD/fancy_gym_amd/libmpk_occ6.so python tools/closed_bench.py 2048 4096 8192 2>&1 | grep -v amdgpu | grep "auto\|lib" | tee gpurun_out/r02g/closed_occ6.md
