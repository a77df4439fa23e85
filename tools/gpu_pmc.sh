#!/bin/bash
# PMC passes (separate runs, --kernel-trace only, never combined with other trace domains) for the warp kernels.
# $1 = cfg "B,H,W,P,rho", $2 = tag.  Writes gpurun_out/pmc_<tag>.{txt,json}
CFG=${1:-128,240,320,128,45}; TAG=${2:-b128}
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/pmc_$TAG; mkdir -p /tmp/pmc_$TAG /root/repo/gpurun_out
i=0
for C in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
         "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU" \
         "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$TAG/p$i -o p --output-format csv -- python /root/repo/tools/pmc_driver.py --cfg $CFG --iters 5 > /tmp/pmc_$TAG/log$i.txt 2>&1 || { echo "pass $i failed: $C"; tail -3 /tmp/pmc_$TAG/log$i.txt; }
done
python /root/repo/tools/pmc_summarize.py /tmp/pmc_$TAG /root/repo/gpurun_out/pmc_$TAG.json > /root/repo/gpurun_out/pmc_$TAG.txt
