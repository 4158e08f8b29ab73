# PMC passes over the DMP quad kernel (cfg3, B = 262144) and, for comparison, the ProDMP stream kernel (cfg2, B = 262144)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "sq:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "act:SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F32" "lds:SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  timeout 120 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $R/gpurun_out/pmcd_${name} -o dmp -- python3 $R/tools/run_cfg.py cfg3 262144 5 > /dev/null 2>&1
  timeout 120 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $R/gpurun_out/pmcd_${name} -o prodmp -- python3 $R/tools/run_cfg.py cfg2 262144 5 > /dev/null 2>&1
done
cd $R && python tools/pmc_summary.py gpurun_out/pmcd_sq gpurun_out/pmcd_act gpurun_out/pmcd_lds
