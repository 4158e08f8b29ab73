cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_fuzz.py 2>&1 | tail -40 > gpurun_out/r02a/pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r02a/bench20.json 2> gpurun_out/r02a/bench20.err
python bench.py > gpurun_out/r02a/bench_default.json 2> gpurun_out/r02a/bench_default.err
MPK_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 50 --warmup 5 --no-cpu > gpurun_out/r02a/bench_g2.json 2> gpurun_out/r02a/bench_g2.err
for B in 2048 8192 65536; do python tools/bench_replan.py $B 50 --graph; python tools/bench_replan.py $B 50; done > gpurun_out/r02a/replan.log 2>&1
nproc > gpurun_out/r02a/nproc.txt; rocm-smi --showclocks --showpower > gpurun_out/r02a/smi.txt 2>&1
tail -5 gpurun_out/r02a/pytest.log
