cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02x
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -8 > gpurun_out/r02x/pytest.log
python -c "import __graft_entry__ as e; e.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -2 > gpurun_out/r02x/smoke.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r02x/bench_k20.json 2> gpurun_out/r02x/bench_k20.err
python bench.py > gpurun_out/r02x/bench_default.json 2> gpurun_out/r02x/bench_default.err
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02x/prof_bench -o bench -- python3 $R/bench.py --no-cpu > $R/gpurun_out/r02x/bench_under_rocprof.json 2> $R/gpurun_out/r02x/bench_under_rocprof.err)
python tools/trace_timed_region.py gpurun_out/r02x/prof_bench/bench_kernel_trace.csv > gpurun_out/r02x/timed_region.txt 2>&1
rm -f gpurun_out/r02x/prof_bench/bench_kernel_trace.csv
for B in 2048 4096 8192 65536; do python tools/bench_replan.py $B 50 --graph; python tools/bench_replan.py $B 50; done 2>&1 | grep -v amdgpu > gpurun_out/r02x/replan.log
python tools/closed_bench.py 2048 4096 8192 16384 65536 2>&1 | grep -v amdgpu > gpurun_out/r02x/closed.md
MPK_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 50 --warmup 5 --no-cpu > gpurun_out/r02x/bench_g2.json 2> gpurun_out/r02x/bench_g2.err
cat gpurun_out/r02x/pytest.log gpurun_out/r02x/smoke.log; tail -c 700 gpurun_out/r02x/bench_k20.json; echo; cat gpurun_out/r02x/timed_region.txt gpurun_out/r02x/replan.log gpurun_out/r02x/closed.md
