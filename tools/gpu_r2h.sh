#!/bin/bash
# Round-2 GPU session H: occupancy variants (8 / 7 waves per SIMD forward, 5 backward).
mkdir -p gpurun_out; cd /root/repo
V=unsuperviseddeephomographyral2018_amd/lib/variants
CFG="128,240,320,128,45;128,480,640,128,64"
: > gpurun_out/r2h_micro.jsonl
timeout 120 python tools/microbench.py --iters 40 --configs "$CFG" 2>/dev/null >> gpurun_out/r2h_micro.jsonl
for n in $1; do
  UH_LIB_PATH=$V/libuh_$n.so timeout 120 python tools/microbench.py --iters 40 --configs "$CFG" 2>/dev/null >> gpurun_out/r2h_micro.jsonl
done
echo done
