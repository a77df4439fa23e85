#!/bin/bash
# Round-2 GPU session F: full GPU tests (after fixes) + per-wave trace of the forward at config 4 and at 240x320.
mkdir -p gpurun_out; cd /root/repo
V=unsuperviseddeephomographyral2018_amd/lib/variants
timeout 900 python -m pytest tests -m gpu -q --tb=short > gpurun_out/r2f_pytest.log 2>&1
UH_LIB_PATH=$V/libuh_trace.so timeout 200 python tools/trace_waves.py 128,480,640,128,64 > gpurun_out/r2f_trace.jsonl 2>gpurun_out/r2f_trace.err
UH_LIB_PATH=$V/libuh_trace.so timeout 200 python tools/trace_waves.py 128,240,320,128,45 >> gpurun_out/r2f_trace.jsonl 2>>gpurun_out/r2f_trace.err
echo done
