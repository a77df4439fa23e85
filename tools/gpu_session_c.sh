#!/bin/bash
# GPU session C: A/B of kernel variants (developer).  $1 = list of variant names ("" = shipped lib); CFG = workloads
# UH_TEST_VARIANT=name additionally runs the warp parity tests against that variant
mkdir -p gpurun_out; cd /root/repo
V=unsuperviseddeephomographyral2018_amd/lib/variants
CFG=${CFG:-"128,240,320,128,45;128,480,640,128,64"}
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3 > gpurun_out/pytest_c.log
if [ -n "$UH_TEST_VARIANT" ]; then UH_LIB_PATH=$V/libuh_$UH_TEST_VARIANT.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "warp or chain or full_size" 2>&1 | tail -3 >> gpurun_out/pytest_c.log; fi
: > gpurun_out/micro_c.log
timeout 120 python tools/microbench.py --iters 50 --configs "$CFG" 2>/dev/null >> gpurun_out/micro_c.log
for n in $1; do
  UH_LIB_PATH=$V/libuh_$n.so timeout 120 python tools/microbench.py --iters 50 --configs "$CFG" 2>/dev/null >> gpurun_out/micro_c.log
done
echo done
