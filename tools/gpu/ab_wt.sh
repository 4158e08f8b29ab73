#!/bin/bash
# A/B of two library builds on the cache-resident (write-through) launches: headline step, closed loop, rollouts
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ab_wt; mkdir -p $O
OLD=$GRAFT_REPO_ROOT/fancy_gym_amd/libmpk_old.so
for r in 1 2; do
  for lib in new old; do
    if [ $lib = old ]; then export MPK_LIB=$OLD; else unset MPK_LIB; fi
    python bench.py --no-cpu --no-streaming --no-overlap 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib K=2000 value %.4g kernel_us %.3f frac %.3f' % (d['value'], d['roofline']['kernel_avg_us'], d['roofline']['frac']))"
    python bench.py --steps 20 --warmup 5 --no-cpu --no-streaming --no-overlap 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib K=20   value %.4g kernel_us %.3f frac %.3f' % (d['value'], d['roofline']['kernel_avg_us'], d['roofline']['frac']))"
  done
done 2>&1 | tee $O/headline.txt
for lib in new old new old; do
  if [ $lib = old ]; then export MPK_LIB=$OLD; else unset MPK_LIB; fi
  echo "== $lib"; python tools/closed_bench.py 4096 8192 2>&1 | grep "auto\|pipe=0 \|duo" 
done 2>&1 | tee $O/closed.txt
for lib in new old; do
  if [ $lib = old ]; then export MPK_LIB=$OLD; else unset MPK_LIB; fi
  echo "== $lib"; python tools/rollout_bench.py 4096 8192 2>&1 | grep "auto"
done 2>&1 | tee $O/rollout.txt
unset MPK_LIB
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
