#!/bin/bash
# GPU session A (round 1): gpu tests, bench line, microbench, rocprofv3 kernel-trace stats of bench.py
mkdir -p gpurun_out; cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -15 > gpurun_out/pytest_a.log
timeout 120 python __graft_entry__.py smoke > gpurun_out/smoke_a.log 2>&1
timeout 600 python bench.py > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err
timeout 300 python tools/microbench.py --iters 100 > gpurun_out/micro_a.log 2> gpurun_out/micro_a.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_a -o bench --output-format csv -- python /root/repo/bench.py --steps 10 --warmup 3 --cpu_baseline 0 > /root/repo/gpurun_out/bench_prof_a.json 2> /root/repo/gpurun_out/bench_prof_a.err )
cp $(find /tmp/prof_a -name "*kernel_stats*" | head -1) gpurun_out/bench_kernel_stats_a.csv 2>/dev/null
nproc > gpurun_out/nproc.txt; lscpu | head -20 >> gpurun_out/nproc.txt
echo done
