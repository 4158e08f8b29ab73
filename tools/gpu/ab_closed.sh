#!/bin/bash
# A/B of two library builds on the closed-loop step and the rollout kernels (alternating, one box)
cd $GRAFT_REPO_ROOT
OLD=$GRAFT_REPO_ROOT/fancy_gym_amd/libmpk_old.so
for lib in new old new old; do
  if [ $lib = old ]; then export MPK_LIB=$OLD; else unset MPK_LIB; fi
  echo "== $lib"; python tools/closed_bench.py 2048 4096 8192 2>&1 | grep "auto \|duo "
  python tools/rollout_bench.py 4096 8192 65536 2>&1 | grep "auto"
done
unset MPK_LIB
