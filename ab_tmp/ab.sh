for rep in 1 2; do
for l in "" ab_tmp/libmpk_old.so; do
  echo "== lib: ${l:-new}"
  MPK_LIB=${l:+$PWD/$l} python tools/bench_replan.py 8192 100 --graph 2>&1 | tail -1 | cut -c1-110
  MPK_LIB=${l:+$PWD/$l} python tools/bench_replan.py 4096 100 --graph 2>&1 | tail -1 | cut -c1-110
  MPK_LIB=${l:+$PWD/$l} python tools/sweep.py 2>/dev/null | grep -i "closed\|dmp" | cut -c1-110
done; done
