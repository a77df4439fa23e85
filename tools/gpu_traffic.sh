#!/bin/bash
# PMC traffic of the warp kernels under bench.py itself (the command the roofline figure is measured on).
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/pmc_bench; mkdir -p /tmp/pmc_bench /root/repo/gpurun_out
CMD="python /root/repo/bench.py --steps 5 --warmup 3 --cpu_baseline 0 --north_star 0"
i=0
for C in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_bench/p$i -o p --output-format csv -- $CMD > /tmp/pmc_bench/log$i.txt 2>&1 || { echo "pass $i failed"; tail -3 /tmp/pmc_bench/log$i.txt; }
done
python /root/repo/tools/traffic_from_pmc.py /tmp/pmc_bench 64 240 320 /root/repo/gpurun_out/traffic_r01.json "bench.py --steps 5 --warmup 3 --cpu_baseline 0 --north_star 0" > /root/repo/gpurun_out/traffic_r01.txt
# kernel-trace stats of the same command (no counters)
rm -rf /tmp/prof_b; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o bench --output-format csv -- python /root/repo/bench.py --steps 30 --warmup 5 --cpu_baseline 0 --north_star 0 > /root/repo/gpurun_out/bench_prof_b.json 2> /root/repo/gpurun_out/bench_prof_b.err
cp $(find /tmp/prof_b -name "*kernel_stats*" | head -1) /root/repo/gpurun_out/bench_kernel_stats_b.csv 2>/dev/null
echo done
