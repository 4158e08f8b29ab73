#!/bin/bash
# matrix-core utilisation of k_traj_wide on the reference's num_basis = 1000 example (tools/wide_bench.py case 1, B = 16384)
TAG=${1:-r03_wide_pmc}
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format json -d $O/p$i -o s -- python3 $R/tools/wide_bench.py 16384 > $O/p$i.out 2> $O/p$i.err
done
cd $R
python tools/pmc_compare.py --skip 2 $O/p* > $O/compare.md 2> $O/compare.err
cat $O/compare.md; tail -3 $O/compare.err
