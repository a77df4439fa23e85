#!/bin/bash
# Round-2 evidence run: GPU tests, smoke, bench lines, rocprofv3 kernel stats (whole bench; config-4 point alone),
# steady-state step breakdown, PMC traffic of the warp kernels (in-step shape and config-4 shape).
mkdir -p gpurun_out; cd /root/repo; export TMPDIR=/tmp
TAG=${1:-r02}
COMMIT=${2:-unknown}
timeout 900 python -m pytest tests -m gpu -q --tb=short > gpurun_out/${TAG}_pytest_full.log 2>&1; tail -5 gpurun_out/${TAG}_pytest_full.log > gpurun_out/${TAG}_pytest.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/${TAG}_smoke.log 2>&1
timeout 900 python bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err
timeout 400 python bench.py --cpu_baseline 0 --north_star 0 --config4 0 --fused_patch 1 > gpurun_out/${TAG}_bench_line_fused_patch.json 2>/dev/null
timeout 400 python bench.py --cpu_baseline 0 --north_star 0 --config4 0 --mid_training_theta 0 > gpurun_out/${TAG}_bench_line_raw_regressor.json 2>/dev/null
# functional check of the N > 1 path of bench.py on a 1-GPU box: 2 ranks share the GPU over gloo (RCCL needs a device per rank)
UH_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 5 --warmup 2 --cpu_baseline 0 > gpurun_out/${TAG}_bench_line_2ranks_gloo_one_gpu.json 2> gpurun_out/${TAG}_bench_2ranks.err
CMD="python /root/repo/bench.py --steps 30 --warmup 5 --cpu_baseline 0 --north_star 0 --config4 0"
( cd /tmp && rm -rf /tmp/prof_$TAG && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o bench --output-format csv -- $CMD > /root/repo/gpurun_out/${TAG}_bench_line_under_rocprof.json 2>/dev/null )
cp $(find /tmp/prof_$TAG -name "*kernel_stats*" | head -1) gpurun_out/${TAG}_bench_kernel_stats.csv 2>/dev/null
python tools/step_breakdown.py $(find /tmp/prof_$TAG -name "*kernel_trace.csv" | head -1) 20 60 > gpurun_out/${TAG}_step_breakdown.txt 2>&1
( cd /tmp && rm -rf /tmp/prof4_$TAG && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof4_$TAG -o c4 --output-format csv -- python /root/repo/bench.py --only_points config4 > /root/repo/gpurun_out/${TAG}_config4_point_under_rocprof.json 2>/dev/null )
cp $(find /tmp/prof4_$TAG -name "*kernel_stats*" | head -1) gpurun_out/${TAG}_config4_kernel_stats.csv 2>/dev/null
for W in bench c4; do
  if [ $W = bench ]; then PC="python /root/repo/bench.py --steps 5 --warmup 3 --cpu_baseline 0 --north_star 0 --config4 0"; DIMS="64 240 320"; else PC="python /root/repo/bench.py --only_points config4"; DIMS="128 480 640"; fi
  ( cd /tmp && rm -rf /tmp/pmc_$W && mkdir -p /tmp/pmc_$W && i=0 && for C in "FETCH_SIZE" "WRITE_SIZE"; do i=$((i+1)); timeout 400 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$W/p$i -o p --output-format csv -- $PC > /tmp/pmc_$W/log$i.txt 2>&1 || echo "pmc pass $W $i failed"; done )
  python tools/traffic_from_pmc.py /tmp/pmc_$W $DIMS gpurun_out/traffic_${TAG}.json "$PC" "measured $(date -u +%Y-%m-%d) on one MI355X at commit $COMMIT by tools/gpu_final_r2.sh" > /dev/null
done
# SQ / TCP / TD / TCC counters of the warp kernels at config 4 (one counter group per pass, --kernel-trace only)
bash tools/gpu_pmc2.sh 128,480,640,128,64 ${TAG}_c4 > gpurun_out/${TAG}_pmc.log 2>&1
echo done
