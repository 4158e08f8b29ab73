# PMC of the closed loop on the ring against k_traj_duo (full step, then replanning step: dispatches in that order per kernel)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS=${RINGC_ARGS:-"65536 4 ring_np=8 ring_ns=1 ring_nc=4"}
for pass in "sq:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "act:SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F32" "lds:SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  timeout 150 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $R/gpurun_out/pmc_ringc_${name} -o ringc -- python3 $R/tools/ringc_pmc_driver.py $ARGS > /dev/null 2>&1
done
cd $R && python tools/pmc_summary.py gpurun_out/pmc_ringc_sq gpurun_out/pmc_ringc_act gpurun_out/pmc_ringc_lds
