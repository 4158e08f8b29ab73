#!/bin/bash
# streaming launch vs trajectory-only vs torch fill under the same counters (write latency, TCP / TA stalls, SQ waits)
TAG=${1:-r03_stream_pmc}
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
            "TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_WRITE_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
            "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_BUSY_CYCLES" \
            "TCC_REQ_sum TCC_WRITE_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_sum" \
            "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_NORMAL_WRITEBACK_sum" \
            "TA_TA_BUSY_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_FLAT_WRITE_WAVEFRONTS_sum" \
            "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format json -d $O/p$i -o s -- python3 $R/tools/stream_pmc_driver.py 262144 3 > $O/p$i.out 2> $O/p$i.err
done
cd $R
python tools/pmc_compare.py --skip 1 $O/p* > $O/compare.md 2> $O/compare.err
cat $O/compare.md; tail -3 $O/compare.err $O/p1.err
