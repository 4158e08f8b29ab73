cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python3 $R/bench.py --no-cpu --no-streaming --no-overlap > $R/gpurun_out/bench_under_rocprof.json 2> $R/gpurun_out/bench_under_rocprof.err
cd $R
python bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench_line.err
python tools/sweep.py > gpurun_out/sweep.md 2> gpurun_out/sweep.err
python tools/rows_probe.py > gpurun_out/rows.md 2>&1
tail -c 600 gpurun_out/bench_line.json; ls gpurun_out/prof_bench
