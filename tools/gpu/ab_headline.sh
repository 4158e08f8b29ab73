#!/bin/bash
# headline kernel (bench.py K = 2000 and K = 20) for several library builds, alternating:  bash tools/gpu/ab_headline.sh lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for lib in "$@"; do
    if [ $lib = - ]; then unset MPK_LIB; else export MPK_LIB=$GRAFT_REPO_ROOT/fancy_gym_amd/$lib; fi
    python bench.py --no-cpu --no-streaming --no-overlap 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-16s K=2000 value %.4g kernel_us %.3f frac %.3f' % ('$lib', d['value'], d['roofline']['kernel_avg_us'], d['roofline']['frac']))"
    python bench.py --steps 20 --warmup 5 --no-cpu --no-streaming --no-overlap 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-16s K=20   value %.4g kernel_us %.3f frac %.3f' % ('$lib', d['value'], d['roofline']['kernel_avg_us'], d['roofline']['frac']))"
  done
done
