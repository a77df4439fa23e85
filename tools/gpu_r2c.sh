#!/bin/bash
# Round-2 GPU session C: parity (sparse backward, dU), structural variants of the forward, PMC of the VGPR-staged variant.
mkdir -p gpurun_out; cd /root/repo
V=unsuperviseddeephomographyral2018_amd/lib/variants
CFG=${CFG:-"128,240,320,128,45;128,480,640,128,64"}
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -25 > gpurun_out/r2c_pytest.log
: > gpurun_out/r2c_micro.jsonl
timeout 120 python tools/microbench.py --iters 50 --configs "$CFG" 2>/dev/null >> gpurun_out/r2c_micro.jsonl
UH_IDENTITY_THETA=1 timeout 120 python tools/microbench.py --iters 50 --configs "128,480,640,128,64" 2>/dev/null >> gpurun_out/r2c_micro.jsonl
for n in $1; do
  UH_LIB_PATH=$V/libuh_$n.so timeout 120 python tools/microbench.py --iters 50 --configs "$CFG" 2>/dev/null >> gpurun_out/r2c_micro.jsonl
done
UH_LIB_PATH=$V/libuh_r01.so UH_IDENTITY_THETA=1 timeout 120 python tools/microbench.py --iters 50 --configs "128,480,640,128,64" 2>/dev/null >> gpurun_out/r2c_micro.jsonl
UH_LIB_PATH=$V/libuh_vgpr.so bash tools/gpu_pmc2.sh 128,480,640,128,64 r2c_vgpr > gpurun_out/r2c_pmc.log 2>&1
echo done
