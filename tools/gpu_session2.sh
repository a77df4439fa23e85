#!/bin/bash
# GPU session 2: full gpu tests, A/B of tile variants, rocprofv3 kernel trace, counter list
mkdir -p gpurun_out; cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -15 > gpurun_out/pytest2.log
V=unsuperviseddeephomographyral2018_amd/lib/variants
for v in "" $V/libuh_rows1.so $V/libuh_rows2.so $V/libuh_rows8.so $V/libuh_noxcd.so; do
  UH_LIB_PATH=$v timeout 120 python tools/microbench.py --iters 100 --configs "128,240,320,128,45;128,480,640,128,64" 2>/dev/null
done > gpurun_out/micro2.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o p2 --output-format csv -- python /root/repo/tools/microbench.py --iters 30 --configs "64,240,320,128,45;128,240,320,128,45" > /dev/null 2>&1 )
find /tmp/prof2 -name "*stats*" | head; cp $(find /tmp/prof2 -name "*kernel_stats*" | head -1) gpurun_out/p2_kernel_stats.csv 2>/dev/null
timeout 60 rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
echo done
