# HBM traffic of the trajectory-only launches k_traj_flat now takes (round 5: two workgroups per CU), from the PMC counters -- separate passes,
# bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 on gfx950 (MI355X_MICROARCH.md, HBM section).   gpurun -- 'bash tools/gpu/flat_pmc.sh [B ...]'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for B in ${@:-65536 262144}; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmcflat_${c} -o b$B -- python3 $R/tools/run_cfg.py cfg2 $B 12 > /dev/null 2>&1
  done
done
cd $R && python tools/pmc_summary.py gpurun_out/pmcflat_FETCH_SIZE gpurun_out/pmcflat_WRITE_SIZE
