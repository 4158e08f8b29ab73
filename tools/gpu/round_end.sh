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
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bench -o bench -- python3 $R/bench.py --no-cpu --no-streaming --no-overlap > $R/$O/bench_under_rocprof.json 2> $R/$O/bench_under_rocprof.err)
python tools/trace_timed_region.py $O/prof_bench/bench_kernel_trace.csv > $O/timed_region.txt 2>&1
gzip -9 -f $O/prof_bench/bench_kernel_trace.csv
bash tools/pmc_traffic.sh 4096 262144 > $O/pmc_traffic.txt 2>&1
python tools/pmc_traffic_json.py gpurun_out/pmct_FETCH_SIZE gpurun_out/pmct_WRITE_SIZE > $O/pmc_traffic.json 2> $O/pmc_traffic.err
python tools/sweep.py > $O/sweep.md 2> $O/sweep.err
python tools/closed_bench.py 2048 4096 8192 16384 65536 2>&1 | grep -v amdgpu > $O/closed.md
python tools/rollout_bench.py 4096 8192 16384 65536 2>&1 | grep -v amdgpu > $O/rollout.md
python tools/phase_bench.py 2048 4096 8192 16384 65536 262144 2>&1 | grep -v amdgpu > $O/phase.md
python tools/dmp_bench.py 2>&1 | grep -v amdgpu > $O/dmp_response.md
python tools/episode_bench.py 2>&1 | grep -v amdgpu > $O/episode_return.md
python tools/wide_bench.py > $O/wide.md 2>&1
for B in 2048 4096 8192 65536; do python tools/bench_replan.py $B 50 --graph; python tools/bench_replan.py $B 50; done 2>&1 | grep -v amdgpu > $O/replan.log
(for B in 1024 4096 8192; do for f in "" "--gate" "--verbose1" "--gate --verbose1"; do python tools/bench_replan.py $B 50 --tt $f --graph; python tools/bench_replan.py $B 50 --tt $f; done; done) 2>&1 | grep -v amdgpu > $O/replan_tt.log
(for B in 1024 4096; do python tools/bench_replan.py $B 50 --tt --graph phase_pipe=0; done) 2>&1 | grep -v amdgpu >> $O/replan_tt.log
# round 6: the learned-phase families through every entry point, the gate's price, chunk sizes, the trajectory-only kernel on the TableTennis shape
python tools/learned_phase_bench.py 1024 2048 4096 8192 65536 2> $O/learned_phase.err | grep -v amdgpu > $O/learned_phase.md
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_lp -o lp -- python3 $R/tools/learned_phase_bench.py 1024 8192 TT-ProDMP BeerPong-ProMP > /dev/null 2>&1)
bash tools/gpu/phase_fused_pmc.sh 8192 65536 > $O/phase_fused_pmc.txt 2>&1
bash tools/gpu/pf_split_sweep.sh > $O/phase_fused_split.md 2>&1
(for o in phase_pipe=0 "" phase_chunk=4 phase_chunk=8; do echo "== $o"; python tools/learned_phase_bench.py 1024 2048 4096 TT-ProDMP BeerPong-ProMP $o 2>/dev/null | grep "closed-loop\|verbose" | cut -d'|' -f2,3,4,5,7; done) > $O/phase_pipe.md 2>&1
python tools/rollout_bench.py 24576 65536 262144 --waves 2>/dev/null | grep LongSimple | cut -d'|' -f2,3,4,5,8 > $O/rollout_waves.md
(for c in 2 4 8; do echo "== phase_chunk=$c"; python tools/learned_phase_bench.py 8192 65536 TT-ProDMP BeerPong-ProMP phase_chunk=$c 2>/dev/null | grep -v "separate launches\|gated\|trajectory |\|^lib\|^| config\|^|---"; done) > $O/phase_fused_chunks.md
(for cfg in "TT-ProDMP" "cfg5 TT-ProMP"; do for B in 8192 65536; do n=40; [ $B -gt 10000 ] && n=10; python3 tools/gate_probe.py "$cfg" $B $n scale=0.2; python3 tools/gate_probe.py "$cfg" $B $n; done; done) 2>&1 | grep -v amdgpu > $O/gate_cost.txt
bash tools/gpu/phase_wpb_sweep.sh 2>&1 | grep -v amdgpu > $O/phase_wpb.md
MPK_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu --no-streaming > $O/bench_k20_rccl1.json 2> $O/bench_k20_rccl1.err
MPK_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 50 --warmup 5 --no-cpu > $O/bench_g2.json 2> $O/bench_g2.err
cat $O/pytest.log $O/smoke.log; tail -c 700 $O/bench_k20.json; echo; cat $O/timed_region.txt; tail -c 900 $O/bench_k20_rccl1.json
