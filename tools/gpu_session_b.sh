#!/bin/bash
# GPU session B: gpu tests, microbench, bench (short)
mkdir -p gpurun_out; cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -x 2>&1 | tail -25 > gpurun_out/pytest_b.log
timeout 300 python tools/microbench.py --iters 100 > gpurun_out/micro_b.log 2> gpurun_out/micro_b.err
timeout 600 python bench.py --cpu_baseline 0 > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err
echo done
