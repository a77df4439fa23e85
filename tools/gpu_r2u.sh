#!/bin/bash
mkdir -p gpurun_out; cd /root/repo
V=unsuperviseddeephomographyral2018_amd/lib/variants
CFG="128,240,320,128,45;128,480,640,128,64"
UH_LIB_PATH=$V/libuh_nw1.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -k "warp_forward or config4_forward or full_size or literal or chain or staged" > gpurun_out/r2u_pytest.log 2>&1
: > gpurun_out/r2u_micro.jsonl
timeout 120 python tools/microbench.py --iters 40 --configs "$CFG" 2>/dev/null >> gpurun_out/r2u_micro.jsonl
for n in $1; do
  UH_LIB_PATH=$V/libuh_$n.so timeout 120 python tools/microbench.py --iters 40 --configs "$CFG" 2>/dev/null >> gpurun_out/r2u_micro.jsonl
  UH_IDENTITY_THETA=1 UH_LIB_PATH=$V/libuh_$n.so timeout 120 python tools/microbench.py --iters 40 --configs "128,480,640,128,64" 2>/dev/null >> gpurun_out/r2u_micro.jsonl
done
timeout 120 python tools/microbench.py --iters 40 --configs "$CFG" 2>/dev/null >> gpurun_out/r2u_micro.jsonl
echo done
