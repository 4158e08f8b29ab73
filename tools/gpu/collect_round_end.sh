#!/bin/bash
# Copies what `tools/gpu/round_end.sh <tag>` left under gpurun_out/<tag>/ into profiles/<tag>_* (run in the build container after the call).
TAG=${1:-r05}
O=gpurun_out/$TAG
P=profiles/${TAG}
cp $O/bench_k20.json ${P}_bench_line.json; cp $O/bench_default.json ${P}_bench_line_default.json
cp $O/bench_g2.json ${P}_bench_line_gloo2.json; cp $O/bench_k20_rccl1.json ${P}_bench_line_rccl1.json
cp $O/prof_bench/bench_kernel_stats.csv ${P}_bench_kernel_stats.csv; cp $O/prof_bench/bench_kernel_trace.csv.gz ${P}_bench_kernel_trace.csv.gz
cp $O/timed_region.txt ${P}_timed_region.txt
cp $O/pmc_traffic.json ${P}_pmc_traffic.json; cp $O/pmc_traffic.json profiles/pmc_traffic.json; cp $O/pmc_traffic.txt ${P}_pmc_traffic.txt
cp $O/closed.md ${P}_closed_loop.md; cp $O/rollout.md ${P}_rollout_tables.md
cp $O/dmp_response.md ${P}_dmp_response.md; cp $O/episode_return.md ${P}_episode_return.md; cp $O/wide.md ${P}_wide.md; cp $O/replan.log ${P}_replan.log
cat $O/pytest.log $O/smoke.log > ${P}_gpu_tests.txt
if [ -f $O/learned_phase.md ]; then      # round 6: <tag>_sweep.md is the learned-phase sweep the review asked for, the BASELINE shapes' table moves beside it
  cp $O/learned_phase.md ${P}_sweep.md; cp $O/sweep.md ${P}_sweep_baseline.md
  { echo "# k_phase_fused: episodes per chunk (phase_chunk) at 8 192 / 65 536 episodes"; echo
    echo '`tools/gpu/round_end.sh`: `tools/learned_phase_bench.py 8192 65536 TT-ProDMP BeerPong-ProMP phase_chunk=..`; us per launch'; echo
    cat $O/phase_fused_chunks.md; echo; echo "(a few thousand episodes: r06_phase_fused_split.md, r06_phase_pipe.md -- the tile split and the producer / consumer form)"; } > ${P}_phase_fused_chunks.md
  { echo "# the validity gate inside the launch: gated against ungated, us per launch (tools/gate_probe.py)"; echo
    echo 'two runs per case; `scale=0.2`: no plan violates (the price of the checks alone), default scale: a few plans violate (`n done`)'; echo; echo '```'; cat $O/gate_cost.txt; echo '```'; } > ${P}_gate_cost.md
  [ -f $O/prof_lp/lp_kernel_stats.csv ] && cp $O/prof_lp/lp_kernel_stats.csv ${P}_learned_phase_kernel_stats.csv
  [ -f $O/phase_fused_pmc.txt ] && { echo "# k_phase_fused: HBM bytes per launch from the PMC counters (tools/gpu/phase_fused_pmc.sh; bytes = (2 FETCH_SIZE + WRITE_SIZE) x 1024)"; echo; echo '```'; cat $O/phase_fused_pmc.txt; echo '```'; } > ${P}_phase_fused_pmc_raw.md
  [ -f $O/phase_fused_split.md ] && { echo "# k_phase_fused<..,act>: tiles of a chunk on several waves (phase_split) x chunk size, us (tools/gpu/pf_split_sweep.sh)"; echo; cat $O/phase_fused_split.md; } > ${P}_phase_fused_split.md
  [ -f $O/phase_pipe.md ] && { echo "# k_phase_fused<..,pipe>: one-wave form (phase_pipe=0) / automatic / chunks of 4 / 8, us"; echo; cat $O/phase_pipe.md; } > ${P}_phase_pipe_raw.md
  [ -f $O/replan_tt.log ] && cp $O/replan_tt.log ${P}_replan_tabletennis.log
  [ -f $O/rollout_waves.md ] && cp $O/rollout_waves.md ${P}_rollout_waves.md
  { echo "# k_phase_fused: waves per workgroup (tiles_wpb) and resident waves per CU (phase_waves), tools/gpu/phase_wpb_sweep.sh"; echo; cat $O/phase_wpb.md; } > ${P}_phase_wpb.md
else
  cp $O/sweep.md ${P}_sweep.md
fi
grep "^|" $O/phase.md > ${P}_per_episode_phase_table.md
python - <<PY
import json
for f in ("${P}_bench_line.json", "${P}_bench_line_default.json", "${P}_bench_line_rccl1.json"):
    d = json.loads(open(f).read().strip().split("\n")[-1]); r = d["roofline"]; rs = d.get("roofline_streaming") or {}
    print(f, "value %.4g" % d["value"], "ms %.5f" % d["ms_per_step"], "kernel_avg_us %.2f" % r["kernel_avg_us"], "frac %.3f" % r["frac"], "traffic", r["traffic"],
          "| streaming frac", rs.get("frac"), rs.get("kernel_avg_us"), rs.get("box_fill_GBps"), "| cpu %.4g" % (d.get("cpu_baseline") or {}).get("value", 0), d.get("kernel_avg_us_max"))
d = json.load(open("${P}_pmc_traffic.json")); print(d["_provenance"][:40], {k: (v["kernel"], v["hbm_bytes_per_launch"]) for k, v in d.items() if k != "_provenance"})
PY
head -3 ${P}_bench_kernel_stats.csv | cut -c1-110; tail -7 ${P}_timed_region.txt; cat ${P}_gpu_tests.txt
