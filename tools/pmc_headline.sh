# PMC of the headline launch (k_traj_tiles<prodmp,act>, cfg2, B = 4096): issue mix, waits and the write path
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "sq:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "act:SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES" "wr:TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum" "tcp:TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TA_BUSY_sum TCP_TA_DATA_STALL_CYCLES_sum"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  timeout 200 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $R/gpurun_out/pmc_head_${name} -o b4096 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu --eager --no-streaming --batch 4096 > /dev/null 2>&1
done
cd $R && python tools/pmc_summary.py $(ls -d gpurun_out/pmc_head_*)
