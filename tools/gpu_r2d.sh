#!/bin/bash
# Round-2 GPU session D: full GPU test suite (new loss-gradient / sparse-backward / tail tests), smoke, variants with path C1.
mkdir -p gpurun_out; cd /root/repo
V=unsuperviseddeephomographyral2018_amd/lib/variants
CFG=${CFG:-"128,240,320,128,45;128,480,640,128,64"}
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r2d_pytest.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/r2d_smoke.log 2>&1
: > gpurun_out/r2d_micro.jsonl
timeout 120 python tools/microbench.py --iters 50 --configs "$CFG" 2>/dev/null >> gpurun_out/r2d_micro.jsonl
UH_IDENTITY_THETA=1 timeout 120 python tools/microbench.py --iters 50 --configs "128,480,640,128,64" 2>/dev/null >> gpurun_out/r2d_micro.jsonl
for n in $1; do
  UH_LIB_PATH=$V/libuh_$n.so timeout 120 python tools/microbench.py --iters 50 --configs "$CFG" 2>/dev/null >> gpurun_out/r2d_micro.jsonl
done
timeout 300 python bench.py --steps 20 --warmup 5 --cpu_baseline 0 > gpurun_out/r2d_bench.log 2>&1
echo done
