#!/bin/bash
mkdir -p gpurun_out; cd /root/repo
timeout 900 python -m pytest tests -m gpu -q --tb=short > gpurun_out/r2l_pytest.log 2>&1
timeout 400 python bench.py --steps 20 --warmup 5 --cpu_baseline 0 --north_star 0 --config4 0 --profile 2 > gpurun_out/r2l_bench_profile2.json 2>/dev/null
echo done
