# PMC passes over the fused replanning step (cfg4, B = 8192: k_traj_quad<prodmp,closed> with one unit per wave)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=${1:-8192}
for pass in "sq:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "act:SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F32" "lds:SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "misc:GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  timeout 120 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $R/gpurun_out/pmcr_${name} -o replan -- python3 $R/tools/bench_replan.py $B 5 > /dev/null 2>&1
done
cd $R && python tools/pmc_summary.py gpurun_out/pmcr_sq gpurun_out/pmcr_act gpurun_out/pmcr_lds gpurun_out/pmcr_misc
