# HBM traffic of the headline kernel from the PMC counters (MI355X_MICROARCH.md, HBM section: separate passes,
# bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 on gfx950).   bash tools/pmc_traffic.sh [B ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for B in ${@:-4096 262144}; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmct_${c} -o b$B -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu --no-streaming --eager --batch $B > /dev/null 2>&1
  done
done
cd $R && python tools/pmc_summary.py gpurun_out/pmct_FETCH_SIZE gpurun_out/pmct_WRITE_SIZE
