#!/bin/bash
# Extended PMC passes (SQ / LDS / TA / TCP / TD stall attribution).  $1 = cfg, $2 = tag, UH_LIB_PATH honoured
CFG=${1:-128,480,640,128,64}; TAG=${2:-big2}
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/pmc_$TAG; mkdir -p /tmp/pmc_$TAG /root/repo/gpurun_out
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU" \
         "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS" \
         "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
         "TD_TD_BUSY_sum TD_TC_STALL_sum TCP_GATE_EN1_sum TCP_TCC_WRITE_REQ_sum" \
         "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum GRBM_GUI_ACTIVE" \
         "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$TAG/p$i -o p --output-format csv -- python /root/repo/tools/pmc_driver.py --cfg $CFG --iters 3 > /tmp/pmc_$TAG/log$i.txt 2>&1 || { echo "pass $i failed: $C"; tail -3 /tmp/pmc_$TAG/log$i.txt; }
done
python /root/repo/tools/pmc_summarize.py /tmp/pmc_$TAG /root/repo/gpurun_out/pmc_$TAG.json > /root/repo/gpurun_out/pmc_$TAG.txt
