#!/bin/bash
# session W: the folded launches (tickets, gather+losses, loss-gradient+warp backward): full GPU suite, bench line, step breakdown
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -x --tb=short > gpurun_out/r2w_tests.txt 2>&1
tail -15 gpurun_out/r2w_tests.txt
timeout 600 python bench.py --cpu_baseline 0 > gpurun_out/r2w_bench.json 2> gpurun_out/r2w_bench.err; tail -2 gpurun_out/r2w_bench.err
CMD="python /root/repo/bench.py --steps 30 --warmup 5 --cpu_baseline 0 --north_star 0 --config4 0"
( cd /tmp && rm -rf /tmp/prof_w && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_w -o bench --output-format csv -- $CMD > /root/repo/gpurun_out/r2w_bench_under_rocprof.json 2>/dev/null )
python tools/step_breakdown.py $(find /tmp/prof_w -name "*kernel_trace.csv" | head -1) 20 60 > gpurun_out/r2w_step_breakdown.txt 2>&1
grep -E "uh::|steps averaged" gpurun_out/r2w_step_breakdown.txt | cut -c1-150
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r2w_bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernels'])
print(d['config4_point']); print(d['north_star_point'])
PY
