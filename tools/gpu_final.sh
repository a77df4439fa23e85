#!/bin/bash
# Round-end evidence run: GPU tests, smoke, bench lines, rocprofv3 kernel stats + steady-state step breakdown, PMC traffic.
mkdir -p gpurun_out; cd /root/repo; export TMPDIR=/tmp
TAG=${1:-f}
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/pytest_$TAG.log
timeout 200 python __graft_entry__.py smoke > gpurun_out/smoke_$TAG.log 2>&1
timeout 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
timeout 400 python bench.py --cpu_baseline 0 --north_star 0 --fused_patch 1 > gpurun_out/bench_${TAG}_fused.json 2>/dev/null
CMD="python /root/repo/bench.py --steps 30 --warmup 5 --cpu_baseline 0 --north_star 0"
( cd /tmp && rm -rf /tmp/prof_$TAG && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o bench --output-format csv -- $CMD > /root/repo/gpurun_out/bench_${TAG}_under_rocprof.json 2>/dev/null )
cp $(find /tmp/prof_$TAG -name "*kernel_stats*" | head -1) gpurun_out/bench_kernel_stats_$TAG.csv 2>/dev/null
python tools/step_breakdown.py $(find /tmp/prof_$TAG -name "*kernel_trace.csv" | head -1) 20 60 > gpurun_out/step_breakdown_$TAG.txt 2>&1
( cd /tmp && rm -rf /tmp/pmc_bench && mkdir -p /tmp/pmc_bench && i=0 && for C in "FETCH_SIZE" "WRITE_SIZE"; do i=$((i+1)); timeout 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_bench/p$i -o p --output-format csv -- python /root/repo/bench.py --steps 5 --warmup 3 --cpu_baseline 0 --north_star 0 > /tmp/pmc_bench/log$i.txt 2>&1 || echo "pmc pass $i failed"; done )
python tools/traffic_from_pmc.py /tmp/pmc_bench 64 240 320 gpurun_out/traffic_$TAG.json "bench.py --steps 5 --warmup 3 --cpu_baseline 0 --north_star 0" > /dev/null
echo done
