#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_mid; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err; tail -c 600 $O/bench_k20.json; echo
python tools/wide_bench.py > $O/wide.md 2>&1; cat $O/wide.md
python tools/closed_bench.py 4096 8192 2>&1 | grep -v amdgpu > $O/closed.md; cat $O/closed.md
for B in 4096 8192; do python tools/bench_replan.py $B 50 --graph; done 2>&1 | grep -v amdgpu > $O/replan.log; cat $O/replan.log
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_replan -o r -- python3 $GRAFT_REPO_ROOT/tools/bench_replan.py 8192 50 --graph > /dev/null 2>&1)
head -12 $O/prof_replan/r_kernel_stats.csv | cut -c1-200
