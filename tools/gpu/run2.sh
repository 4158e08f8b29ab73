cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
python -m pytest tests/test_gpu_blackbox.py tests/test_gpu_switches.py tests/test_gpu_ode.py tests/test_gpu_edge_cases.py -m gpu -q -x 2>&1 | tail -30 > gpurun_out/r02b/pytest.log
tail -3 gpurun_out/r02b/pytest.log
python tools/closed_bench.py > gpurun_out/r02b/closed_occ7.md 2>&1
MPK_LIB=$PWD/fancy_gym_amd/libmpk_occ4.so python tools/closed_bench.py > gpurun_out/r02b/closed_occ4.md 2>&1
for B in 2048 8192; do python tools/bench_replan.py $B 50 --graph; done > gpurun_out/r02b/replan.log 2>&1
cat gpurun_out/r02b/closed_occ7.md gpurun_out/r02b/closed_occ4.md | grep -v amdgpu.ids
