cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02f
for v in "" _t6 _t5 _s6 _s5; do
  for rep in 1 2; do
    MPK_LIB=$PWD/fancy_gym_amd/libmpk$v.so python bench.py --steps 2000 --warmup 100 --no-cpu --no-streaming 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib$v', 'ms_per_step', round(d['ms_per_step']*1e3,3), 'us  kernel_avg', round(d['roofline']['kernel_avg_us'],3), 'us', d['roofline']['kernel'])"
  done
done | tee gpurun_out/r02f/tiles_occ.log
