#!/bin/bash
# Regenerates the round's bench evidence on the GPU box:  gpurun --timeout 2400 -- 'bash tools/gpu/round_end.sh r03'
#   gpurun_out/<tag>/{pytest.log, smoke.log, bench_k20.json, bench_default.json, prof_bench/*, timed_region.txt, ...}
# The kernel trace the per-phase table is derived from is KEPT (gzip) beside the --stats CSVs.
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -8 > $O/pytest.log
python -c "import __graft_entry__ as e; e.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -2 > $O/smoke.log
python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bench -o bench -- python3 $R/bench.py --no-cpu --no-streaming > $R/$O/bench_under_rocprof.json 2> $R/$O/bench_under_rocprof.err)
python tools/trace_timed_region.py $O/prof_bench/bench_kernel_trace.csv > $O/timed_region.txt 2>&1
gzip -9 -f $O/prof_bench/bench_kernel_trace.csv
MPK_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu --no-streaming > $O/bench_k20_rccl1.json 2> $O/bench_k20_rccl1.err
MPK_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 50 --warmup 5 --no-cpu > $O/bench_g2.json 2> $O/bench_g2.err
cat $O/pytest.log $O/smoke.log; tail -c 700 $O/bench_k20.json; echo; cat $O/timed_region.txt; tail -c 900 $O/bench_k20_rccl1.json
