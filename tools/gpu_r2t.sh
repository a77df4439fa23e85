#!/bin/bash
# Round-2 GPU session T: parity + timing after the row-block DMA staging and backward VALU trims.
mkdir -p gpurun_out; cd /root/repo
timeout 900 python -m pytest tests -m gpu -q --tb=short > gpurun_out/r2t_pytest.log 2>&1
: > gpurun_out/r2t_micro.jsonl
for i in 1 2; do
timeout 120 python tools/microbench.py --iters 40 --configs "128,240,320,128,45;128,480,640,128,64" 2>/dev/null >> gpurun_out/r2t_micro.jsonl
done
UH_IDENTITY_THETA=1 timeout 120 python tools/microbench.py --iters 40 --configs "128,480,640,128,64" 2>/dev/null >> gpurun_out/r2t_micro.jsonl
echo done
