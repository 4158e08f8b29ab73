#!/bin/bash
# PMC passes of the producer / consumer forms at 1 024 / 4 096 episodes (the kernels episode_pmc.sh measured in their one-wave form):
# gpurun -- 'bash tools/gpu/pipe_pmc.sh'  -> gpurun_out/r06_pipe_pmc/summary.md (appended to profiles/r06_episode_pmc.md by hand)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_pipe_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for which in roll roll_rw tt_lean tt_step; do
  for B in 1024 4096; do
    i=0
    for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
                "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_MFMA"; do
      i=$((i+1))
      timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format json -d $O/${which}_${B}_p$i -o s -- python3 $R/tools/episode_pmc_driver.py $which $B 30 > $O/${which}_${B}_p$i.out 2> $O/${which}_${B}_p$i.err
    done
  done
done
cd $R
python tools/episode_pmc_summary.py $O > $O/summary.md 2> $O/summary.err
cat $O/summary.md; tail -3 $O/summary.err
