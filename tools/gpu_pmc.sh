#!/bin/bash
# PMC passes (separate runs, --kernel-trace only) for the warp kernels.  $1 = cfg, $2 = tag
CFG=${1:-128,240,320,128,45}; TAG=${2:-b128}
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/pmc_$TAG; mkdir -p /tmp/pmc_$TAG
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU" \
         "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
         "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$TAG/p$i -o p --output-format csv -- python /root/repo/tools/pmc_driver.py --cfg $CFG --iters 5 > /tmp/pmc_$TAG/log$i.txt 2>&1 || echo "pass $i failed: $C"
done
python /root/repo/tools/pmc_summarize.py /tmp/pmc_$TAG > /root/repo/gpurun_out/pmc_$TAG.txt
