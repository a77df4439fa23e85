#!/bin/bash
# session X: quick check of the folded tail: the tests that touch it, then the steady-state step breakdown
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -q -m gpu -x --tb=short -k "gather or ticket or loss or chain or tail or patch" > gpurun_out/r2x_tests.txt 2>&1
tail -3 gpurun_out/r2x_tests.txt
CMD="python /root/repo/bench.py --steps 30 --warmup 5 --cpu_baseline 0 --north_star 0 --config4 0"
( cd /tmp && rm -rf /tmp/prof_x && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o bench --output-format csv -- $CMD > /root/repo/gpurun_out/r2x_bench_under_rocprof.json 2>/dev/null )
python tools/step_breakdown.py $(find /tmp/prof_x -name "*kernel_trace.csv" | head -1) 20 60 > gpurun_out/r2x_step_breakdown.txt 2>&1
grep -E "uh::|steps averaged" gpurun_out/r2x_step_breakdown.txt | grep -v bias | cut -c1-150
