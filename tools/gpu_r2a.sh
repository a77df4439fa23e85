#!/bin/bash
# Round-2 GPU session A: parity of the three-path warp kernels, float4 copy yardstick, A/B of variants.
mkdir -p gpurun_out; cd /root/repo
V=unsuperviseddeephomographyral2018_amd/lib/variants
CFG=${CFG:-"128,240,320,128,45;128,480,640,128,64"}
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r2a_pytest.log
timeout 120 tools/ubench/copybw > gpurun_out/r2a_copybw.jsonl 2>&1
: > gpurun_out/r2a_micro.jsonl
timeout 120 python tools/microbench.py --iters 50 --configs "$CFG" 2>/dev/null >> gpurun_out/r2a_micro.jsonl
for n in $1; do
  UH_LIB_PATH=$V/libuh_$n.so timeout 120 python tools/microbench.py --iters 50 --configs "$CFG" 2>/dev/null >> gpurun_out/r2a_micro.jsonl
done
echo done
