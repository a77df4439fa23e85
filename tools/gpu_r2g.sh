#!/bin/bash
# Round-2 GPU session G: wave traces at two occupancies (is the forward issue-bound or latency-bound?), bench line with config4_point.
mkdir -p gpurun_out; cd /root/repo
V=unsuperviseddeephomographyral2018_amd/lib/variants
: > gpurun_out/r2g_trace.jsonl
for n in trace trace_l12; do
  UH_LIB_PATH=$V/libuh_$n.so timeout 200 python tools/trace_waves.py 128,480,640,128,64 >> gpurun_out/r2g_trace.jsonl 2>/dev/null
  UH_IDENTITY_THETA=1 UH_LIB_PATH=$V/libuh_$n.so timeout 200 python tools/trace_waves.py 128,480,640,128,64 >> gpurun_out/r2g_trace.jsonl 2>/dev/null
  UH_LIB_PATH=$V/libuh_$n.so timeout 120 python tools/microbench.py --iters 30 --configs "128,480,640,128,64" 2>/dev/null >> gpurun_out/r2g_micro.jsonl
done
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r2g_bench.log 2>&1
timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -k "hipgraph" > gpurun_out/r2g_pytest.log 2>&1
echo done
