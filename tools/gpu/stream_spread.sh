#!/bin/bash
# streaming-row experiments (skew sweep, 2-vs-3 arrays, sysfs timeline) + a per-channel look at the L2->fabric write
# requests of the streaming launch.   gpurun --timeout 1500 -- 'bash tools/gpu/stream_spread.sh r03_spread'
TAG=${1:-r03_spread}
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-cpu > $O/bench_k20.json 2> $O/bench_k20.err
python tools/skew_probe.py 262144 3 > $O/skew.md 2> $O/skew.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
for c in TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_RDREQ; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format json csv -d $O/pmc_$c -o s -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-streaming --eager --batch 262144 > /dev/null 2> $O/pmc_$c.err
done
cd $R
python - <<'PY' > $O/pmc_channels.md 2>&1
import glob, json, os, sys, collections
O = os.environ.get("O") or sys.argv[-1]
PY
ls -la $O $O/pmc_TCC_EA0_WRREQ 2>/dev/null | head -40
cat $O/skew.md; tail -5 $O/skew.err
grep -n "TCC_EA0_WRREQ\|TCC_EA0_RDREQ" $O/counters_list.txt | head -20
