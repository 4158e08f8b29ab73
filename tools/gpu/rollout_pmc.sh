#!/bin/bash
TAG=${1:-r03_rollout_pmc}
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVES SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format json -d $O/p$i -o s -- python3 $R/tools/rollout_pmc_driver.py 4096 30 > $O/p$i.out 2> $O/p$i.err
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format json -d $O/q$i -o s -- python3 $R/tools/rollout_pmc_driver.py 4096 30 pd_quad=2 > $O/q$i.out 2> $O/q$i.err
done
cd $R
python tools/pmc_compare.py --skip 5 $O/p* > $O/compare_ng1.md 2>&1; python tools/pmc_compare.py --skip 5 $O/q* > $O/compare_ng4.md 2>&1
cat $O/compare_ng1.md $O/compare_ng4.md
