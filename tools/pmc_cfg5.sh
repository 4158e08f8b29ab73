# PMC: tile-major kernel on cfg5 (ProMP 7x3x350 + actions, B = 1024, T*D = 2 mod 4) next to cfg2 (B = 4096)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "sq:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "lds:SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "misc:GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  timeout 120 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $R/gpurun_out/pmc5_${name} -o cfg5 -- python3 $R/tools/run_cfg.py cfg5act 1024 8 > /dev/null 2>&1
  timeout 120 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $R/gpurun_out/pmc5_${name} -o cfg2 -- python3 $R/tools/run_cfg.py cfg2act 4096 8 > /dev/null 2>&1
done
cd $R && python tools/pmc_summary.py gpurun_out/pmc5_sq gpurun_out/pmc5_lds gpurun_out/pmc5_misc
