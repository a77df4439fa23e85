#!/bin/bash
# Round-2 GPU session B: parity again (lerp-form backward), PMC of the three-path kernels at config 4, cache-policy variants.
mkdir -p gpurun_out; cd /root/repo
V=unsuperviseddeephomographyral2018_amd/lib/variants
CFG=${CFG:-"128,240,320,128,45;128,480,640,128,64"}
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r2b_pytest.log
: > gpurun_out/r2b_micro.jsonl
timeout 120 python tools/microbench.py --iters 50 --configs "$CFG" 2>/dev/null >> gpurun_out/r2b_micro.jsonl
for n in $1; do
  UH_LIB_PATH=$V/libuh_$n.so timeout 120 python tools/microbench.py --iters 50 --configs "$CFG" 2>/dev/null >> gpurun_out/r2b_micro.jsonl
done
bash tools/gpu_pmc2.sh 128,480,640,128,64 r2b_c4 > gpurun_out/r2b_pmc.log 2>&1
echo done
