cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02k
python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_switches.py -m gpu -q 2>&1 | grep -E "passed|failed|Error|error|assert|FAILED" | tail -8 | tee gpurun_out/r02k/pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r02k/bench_k20.json 2> gpurun_out/r02k/bench_k20.err
python bench.py > gpurun_out/r02k/bench_default.json 2> gpurun_out/r02k/bench_default.err
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02k/prof_bench -o bench -- python3 $R/bench.py --no-cpu > $R/gpurun_out/r02k/bench_under_rocprof.json 2> $R/gpurun_out/r02k/bench_under_rocprof.err)
python tools/trace_timed_region.py gpurun_out/r02k/prof_bench/bench_kernel_trace.csv > gpurun_out/r02k/timed_region.txt 2>&1
(cd /tmp && export TMPDIR=/tmp; for B in 4096 262144; do for c in FETCH_SIZE WRITE_SIZE; do timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/r02k/pmct_${c} -o b$B -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu --no-streaming --eager --batch $B > /dev/null 2>&1; done; done)
python tools/pmc_summary.py gpurun_out/r02k/pmct_FETCH_SIZE gpurun_out/r02k/pmct_WRITE_SIZE > gpurun_out/r02k/pmc_traffic.txt 2>&1
python tools/clock_probe.py 4 2>&1 | grep -v amdgpu > gpurun_out/r02k/clock_probe.md
python tools/phase_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r02k/phase.md
nproc > gpurun_out/r02k/nproc.txt
find gpurun_out/r02k/prof_bench -name "*kernel_trace.csv" -size +3M -delete
tail -c 900 gpurun_out/r02k/bench_k20.json; echo; cat gpurun_out/r02k/pmc_traffic.txt gpurun_out/r02k/clock_probe.md gpurun_out/r02k/phase.md gpurun_out/r02k/timed_region.txt
