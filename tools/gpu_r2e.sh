#!/bin/bash
# Round-2 GPU session E: failing tests with full tracebacks; warp time per theta family (where does config 4 lose time?).
mkdir -p gpurun_out; cd /root/repo
timeout 900 python -m pytest tests -m gpu -q --tb=long -k "call_surface or dU_large or patch_backward_equals or tail_plan" > gpurun_out/r2e_pytest.log 2>&1
: > gpurun_out/r2e_micro.jsonl
for k in shift rot5 rot15 rot45 zoomin zoomout zoomout2 persp; do
  UH_THETA_KIND=$k timeout 120 python tools/microbench.py --iters 30 --configs "128,480,640,128,64" 2>/dev/null >> gpurun_out/r2e_micro.jsonl
done
echo done
