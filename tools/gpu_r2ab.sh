#!/bin/bash
# session AB: bias-gradient finish with 16 loads in flight: epilogue tests, then the step breakdown
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_model.py -q -m gpu -x --tb=short -k "epilogue or overfits or end_to_end or hipgraph" > gpurun_out/r2ab_tests.txt 2>&1
tail -2 gpurun_out/r2ab_tests.txt
CMD="python /root/repo/bench.py --steps 30 --warmup 5 --cpu_baseline 0 --north_star 0 --config4 0"
( cd /tmp && rm -rf /tmp/prof_ab && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -o bench --output-format csv -- $CMD > /root/repo/gpurun_out/r2ab_bench_under_rocprof.json 2>/dev/null )
python tools/step_breakdown.py $(find /tmp/prof_ab -name "*kernel_trace.csv" | head -1) 20 60 > gpurun_out/r2ab_step_breakdown.txt 2>&1
grep -E "uh::|steps averaged" gpurun_out/r2ab_step_breakdown.txt | cut -c1-150
timeout 300 python bench.py --cpu_baseline 0 --north_star 0 --config4 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
