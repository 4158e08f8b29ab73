#!/bin/bash
# SQ_INSTS_VALU per episode of the per-episode ProDMP kernel, per-episode rounds vs flat rounds (cfg2' = cfg2 + learned tau)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_phase_flat_pmc; mkdir -p $O
for B in 16384 65536; do
for flat in 0 1; do
for pass in "sq:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "lds:SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  timeout 120 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/${name}_B${B}_flat${flat} -o cfg2tau -- python3 $R/tools/run_cfg.py cfg2tau $B 5 phase_flat=$flat > /dev/null 2>&1
done; done; done
cd $R && python tools/pmc_summary.py $O/* > $O/summary.txt 2>&1; cat $O/summary.txt
