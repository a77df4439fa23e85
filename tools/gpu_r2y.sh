#!/bin/bash
# session Y: step breakdown with the last-block tickets on (default) and off (UH_TICKETS=0: separate finish kernels)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -q -m gpu -x --tb=short -k "gather or ticket or loss or chain or tail or patch" > gpurun_out/r2y_tests.txt 2>&1
tail -3 gpurun_out/r2y_tests.txt
for T in 1 0; do
CMD="python /root/repo/bench.py --steps 30 --warmup 5 --cpu_baseline 0 --north_star 0 --config4 0"
( cd /tmp && rm -rf /tmp/prof_y$T && UH_TICKETS=$T timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_y$T -o bench --output-format csv -- $CMD > /root/repo/gpurun_out/r2y_bench_under_rocprof_t$T.json 2>/dev/null )
python tools/step_breakdown.py $(find /tmp/prof_y$T -name "*kernel_trace.csv" | head -1) 20 60 > gpurun_out/r2y_step_breakdown_t$T.txt 2>&1
echo "== UH_TICKETS=$T"; grep -E "uh::|steps averaged" gpurun_out/r2y_step_breakdown_t$T.txt | grep -v bias | cut -c1-150
done
