# PMC of the producer / consumer closed-loop kernel (k_traj_pipe, automatic selection) at a few thousand episodes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=${1:-4096}
for pass in "sq:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "act:SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "lds:SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  timeout 120 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $R/gpurun_out/pmc_pipe_${name} -o closed -- python3 $R/tools/run_cfg.py cfg2closed $B 40 > /dev/null 2>&1
done
cd $R && python tools/pmc_summary.py $(ls -d gpurun_out/pmc_pipe_*)
