# HBM traffic of the one-launch entry points of the learned-phase families (k_phase_fused) from the PMC counters -- separate passes,
# bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 on gfx950 (MI355X_MICROARCH.md, HBM section); algorithmic bytes per episode: TableTennis-ProDMP
# 29 660 (actions) / 29 772 (closed loop), BeerPong-ProMP 25 428 / 25 540.   gpurun -- 'bash tools/gpu/phase_fused_pmc.sh [B ...]'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for B in ${@:-8192 65536}; do
  for cfg in ttprodmpact ttprodmpclosed beerpongact; do
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmcpf_${c} -o ${cfg}_b$B -- python3 $R/tools/run_cfg.py $cfg $B 12 > /dev/null 2>&1
    done
  done
done
cd $R && python tools/pmc_summary.py gpurun_out/pmcpf_FETCH_SIZE gpurun_out/pmcpf_WRITE_SIZE
