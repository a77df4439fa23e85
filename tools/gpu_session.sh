#!/bin/bash
# ONE parametrised GPU session script (replaces the per-session gpu_r2*.sh / gpu_final*.sh one-offs).
#   gpurun --timeout 1500 -- 'bash tools/gpu_session.sh TAG stage [stage ...]'
# Every stage writes gpurun_out/${TAG}_*; copy what is to be judged into profiles/ by hand.  Stages:
#   tests[:K]       pytest -m gpu (optionally -k K), full log + tail
#   smoke           __graft_entry__.smoke()
#   bench           default bench line (+ fused-patch and raw-regressor lines with bench_extra)
#   bench_extra     the two extra lines
#   other           BASELINE configs[4] (supervised, batch 256) and the reference's default batch 128
#   bench2          python bench.py --gpus 2 on ONE GPU over gloo (functional check of the N > 1 path and the self-launch)
#   rocprof         rocprofv3 --kernel-trace --stats of the bench command + steady-state step breakdown
#   rocprof_c4      the same of bench.py --only_points config4 (one shape per kernel name)
#   traffic         FETCH_SIZE / WRITE_SIZE passes (own runs, --kernel-trace only) -> traffic_${TAG}.json, both shapes
#   pmc_c4          SQ / LDS / TA / TCP / TCC counter groups at config 4 (tools/gpu_pmc2.sh)
#   micro:V1,V2     A/B/A/B microbench of the shipped library against lib/variants/libuh_V*.so (tools/variants.sh builds them)
#   vtests:V[:K]    pytest -m gpu with UH_LIB_PATH = variant V
#   inputs          section-8 f3 timing (tools/time_inputs.py)
#   dp              tests/test_gpu_dp_product.py (2 gloo ranks on the one GPU: product DP step == one tower)
#   bench8[:N]      python bench.py --gpus N (default 8) on ONE GPU over gloo, find pass staggered and not, cold MIOpen db each
#   cold            tools/cold_forward.py (in-step forward: cold / warm x batch sweep x theta law)
#   train_ref       tools/train_reference_schedule.sh (150 000 steps at the reference's schedule + test loop)
#   dpsmooth / dpnoise   the two data-dependent DP tests on smooth textures (pinned / default solvers) / on the default solvers
#   coldv:V1,V2     cold-forward A/B/A/B of the shipped library against variants (batches 64, 128)
#   power:LIBS:K    tools/power_ab.py: duration + socket power + energy per launch of kernel K (fwd | bwd) per library build
#   trace:V         per-wave phase trace with the -DUH_WARP_TRACE variant V (tools/trace_waves.py)
#   anatomy:V       what the in-step forward's fixed cost is made of (tools/trace_launch_anatomy.py, trace variant V): cold and warm
#   ablate[:STEPS]  tools/train_from_disk.py, four arms (JPEG / PNG files x joint augmentation 0.5 / off), each tested with and
#                   without the reference's disjoint test augmentation
#   train_disk[:STEPS:PAIRS]  the reference's schedule (150 000 steps) FROM JPEG FILES with augmentation 0.5 (tools/train_from_disk.py)
#   conv_relu_probe aten::miopen_convolution_relu per conv shape against conv + the shipped bias/ReLU pass (tools/conv_relu_probe.py)
#   cpu_threads     the CPU leg alone at 8 / 32 / 128 / 256 host threads
mkdir -p gpurun_out; cd /root/repo; export TMPDIR=/tmp
TAG=$1; shift
COMMIT=${UH_COMMIT:-unknown}
VDIR=unsuperviseddeephomographyral2018_amd/lib/variants
MCFG=${MCFG:-"128,240,320,128,45;128,480,640,128,64"}
BENCH="python /root/repo/bench.py --steps 30 --warmup 5 --cpu_baseline 0 --north_star 0 --config4 0 --quality 0 --traffic 0"
for ST in "$@"; do
  IFS=: read -r NAME A1 A2 <<< "$ST"
  echo "== stage $NAME $A1 $A2"
  case $NAME in
    tests)  timeout 1200 python -m pytest tests -m gpu -q --tb=short ${A1:+-k "$A1"} > gpurun_out/${TAG}_pytest_full.log 2>&1; tail -5 gpurun_out/${TAG}_pytest_full.log | tee gpurun_out/${TAG}_pytest.log ;;
    smoke)  timeout 300 python __graft_entry__.py smoke > gpurun_out/${TAG}_smoke.log 2>&1; tail -2 gpurun_out/${TAG}_smoke.log ;;
    bench)  timeout 1200 python bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err; cut -c1-600 gpurun_out/${TAG}_bench_line.json ;;
    bench_extra)
            timeout 400 python bench.py --cpu_baseline 0 --north_star 0 --config4 0 --quality 0 --fused_patch 1 > gpurun_out/${TAG}_bench_line_fused_patch.json 2>/dev/null
            timeout 400 python bench.py --cpu_baseline 0 --north_star 0 --config4 0 --quality 0 --mid_training_theta 0 > gpurun_out/${TAG}_bench_line_raw_regressor.json 2>/dev/null ;;
    other)  ( echo "== config 5: supervised 4-pt (reference flag h_loss), batch 256"; timeout 600 python bench.py --loss_type h_loss --per_gpu_batch 256 --cpu_baseline 0 --north_star 0 --config4 0 --quality 0 2>/dev/null | cut -c1-900
              echo "== the reference's default batch 128, photometric l1"; timeout 600 python bench.py --per_gpu_batch 128 --cpu_baseline 0 --north_star 0 --config4 0 --quality 0 2>/dev/null | cut -c1-900 ) > gpurun_out/${TAG}_other_configs.txt; cut -c1-300 gpurun_out/${TAG}_other_configs.txt ;;
    bench2) UH_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --cpu_baseline 0 --quality 0 > gpurun_out/${TAG}_bench_line_2ranks_gloo_one_gpu.json 2> gpurun_out/${TAG}_bench_2ranks.err; cut -c1-400 gpurun_out/${TAG}_bench_line_2ranks_gloo_one_gpu.json ;;
    rocprof)
            ( cd /tmp && rm -rf /tmp/prof_$TAG && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o bench --output-format csv -- $BENCH > /root/repo/gpurun_out/${TAG}_bench_line_under_rocprof.json 2>/dev/null )
            cp $(find /tmp/prof_$TAG -name "*kernel_stats*" | head -1) gpurun_out/${TAG}_bench_kernel_stats.csv 2>/dev/null
            python tools/step_breakdown.py $(find /tmp/prof_$TAG -name "*kernel_trace.csv" | head -1) 5 30 60 > gpurun_out/${TAG}_step_breakdown.txt 2>&1
            python tools/timed_steps_stats.py $(find /tmp/prof_$TAG -name "*kernel_trace.csv" | head -1) 5 30 > gpurun_out/${TAG}_bench_kernel_stats_timed_steps.csv 2> gpurun_out/${TAG}_timed_steps.err
            grep -E "warp_|Block" gpurun_out/${TAG}_bench_kernel_stats_timed_steps.csv | cut -c1-260 ;;
    rocprof_c4)
            ( cd /tmp && rm -rf /tmp/prof4_$TAG && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof4_$TAG -o c4 --output-format csv -- python /root/repo/bench.py --only_points config4 > /root/repo/gpurun_out/${TAG}_config4_point_under_rocprof.json 2>/dev/null )
            cp $(find /tmp/prof4_$TAG -name "*kernel_stats*" | head -1) gpurun_out/${TAG}_config4_kernel_stats.csv 2>/dev/null
            grep -E "warp_(forward|backward)_kernel" gpurun_out/${TAG}_config4_kernel_stats.csv | cut -c1-200 ;;
    traffic)
            for W in bench c4; do
              if [ $W = bench ]; then PC="python /root/repo/bench.py --steps 5 --warmup 3 --cpu_baseline 0 --north_star 0 --config4 0 --quality 0 --traffic 0 --profile 0"; DIMS="64 240 320"; else PC="python /root/repo/bench.py --only_points config4"; DIMS="128 480 640"; fi
              ( cd /tmp && rm -rf /tmp/pmc_$W && mkdir -p /tmp/pmc_$W && i=0 && for C in "FETCH_SIZE" "WRITE_SIZE"; do i=$((i+1)); timeout 400 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$W/p$i -o p --output-format csv -- $PC > /tmp/pmc_$W/log$i.txt 2>&1 || echo "pmc pass $W $i failed"; done )
              python tools/traffic_from_pmc.py /tmp/pmc_$W $DIMS gpurun_out/traffic_${TAG}.json "$PC" "measured $(date -u +%Y-%m-%d) on one MI355X at commit $COMMIT by tools/gpu_session.sh traffic" "$(cat unsuperviseddeephomographyral2018_amd/lib/libuh_hotpath.so.sha256)" > /dev/null
            done ;;
    pmc_c4) bash tools/gpu_pmc2.sh 128,480,640,128,64 ${TAG}_c4 > gpurun_out/${TAG}_pmc.log 2>&1 ;;
    micro)  : > gpurun_out/${TAG}_micro.jsonl
            for rep in 1 2; do
              timeout 150 python tools/microbench.py --iters 40 --configs "$MCFG" 2>/dev/null >> gpurun_out/${TAG}_micro.jsonl
              for V in ${A1//,/ }; do UH_LIB_PATH=$VDIR/libuh_$V.so timeout 150 python tools/microbench.py --iters 40 --configs "$MCFG" 2>/dev/null >> gpurun_out/${TAG}_micro.jsonl; done
            done
            python tools/show_micro.py gpurun_out/${TAG}_micro.jsonl 2>/dev/null | tail -40 ;;
    vtests) UH_LIB_PATH=$VDIR/libuh_$A1.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_model.py -m gpu -q --tb=short -x ${A2:+-k "$A2"} > gpurun_out/${TAG}_pytest_$A1.log 2>&1; tail -3 gpurun_out/${TAG}_pytest_$A1.log ;;
    inputs) timeout 900 python tools/time_inputs.py > gpurun_out/${TAG}_inputs.txt 2>&1; tail -30 gpurun_out/${TAG}_inputs.txt
            ( cd /tmp && rm -rf /tmp/profi_$TAG && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/profi_$TAG -o inp --output-format csv -- python /root/repo/tools/time_inputs.py --kernel_only 1 > /dev/null 2>&1 )
            cp $(find /tmp/profi_$TAG -name "*kernel_stats*" | head -1) gpurun_out/${TAG}_inputs_kernel_stats.csv 2>/dev/null
            grep prepare_inputs gpurun_out/${TAG}_inputs_kernel_stats.csv | cut -c1-220 ;;
    trace)  UH_LIB_PATH=$VDIR/libuh_$A1.so timeout 300 python tools/trace_waves.py > gpurun_out/${TAG}_wave_trace_$A1.jsonl 2>gpurun_out/${TAG}_wave_trace.err
            UH_TRACE_BWD=1 UH_LIB_PATH=$VDIR/libuh_$A1.so timeout 300 python tools/trace_waves.py >> gpurun_out/${TAG}_wave_trace_$A1.jsonl 2>>gpurun_out/${TAG}_wave_trace.err
            cut -c1-1500 gpurun_out/${TAG}_wave_trace_$A1.jsonl ;;
    anatomy) : > gpurun_out/${TAG}_launch_anatomy.jsonl
            for WARM in 0 1; do UH_LIB_PATH=$VDIR/libuh_$A1.so timeout 300 python tools/trace_launch_anatomy.py --reps 5 --warm $WARM >> gpurun_out/${TAG}_launch_anatomy.jsonl 2>> gpurun_out/${TAG}_launch_anatomy.err; done
            UH_LIB_PATH=$VDIR/libuh_$A1.so timeout 300 python tools/trace_launch_anatomy.py --reps 3 --shape 128,240,320,128,45 >> gpurun_out/${TAG}_launch_anatomy.jsonl 2>> gpurun_out/${TAG}_launch_anatomy.err
            UH_LIB_PATH=$VDIR/libuh_$A1.so timeout 300 python tools/trace_launch_anatomy.py --reps 3 --shape 32,240,320,128,45 >> gpurun_out/${TAG}_launch_anatomy.jsonl 2>> gpurun_out/${TAG}_launch_anatomy.err
            grep summary gpurun_out/${TAG}_launch_anatomy.jsonl | cut -c1-700; tail -3 gpurun_out/${TAG}_launch_anatomy.err ;;
    ablate) timeout 1500 python tools/train_from_disk.py --arms jpg:0.5,jpg:0,png:0.5,png:0 --test_do_augment 0.5,0 --steps ${A1:-8000} --train_pairs 8192 --log_every 2000 > gpurun_out/${TAG}_train_from_disk_ablation.txt 2> gpurun_out/${TAG}_train_from_disk_ablation.err
            grep -E "^\||dataset|trained|RESULT|free" gpurun_out/${TAG}_train_from_disk_ablation.txt | cut -c1-400; tail -3 gpurun_out/${TAG}_train_from_disk_ablation.err ;;
    train_disk) timeout ${TRAIN_TIMEOUT:-1700} python tools/train_from_disk.py --arms jpg:0.5 --test_do_augment 0.5,0 --steps ${A1:-150000} --train_pairs ${A2:-65536} --log_every 10000 > gpurun_out/${TAG}_train_from_disk_reference_schedule.txt 2> gpurun_out/${TAG}_train_from_disk_reference_schedule.err
            grep -E "^\||dataset|trained|RESULT|Average|ercentile|per-pair" gpurun_out/${TAG}_train_from_disk_reference_schedule.txt | cut -c1-500; grep "Train: step" gpurun_out/${TAG}_train_from_disk_reference_schedule.txt | tail -3 | cut -c1-300; tail -3 gpurun_out/${TAG}_train_from_disk_reference_schedule.err ;;
    conv_relu_probe) ( cd /tmp && rm -rf /tmp/pcr_$TAG && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pcr_$TAG -o cr --output-format csv -- python /root/repo/tools/conv_relu_probe.py > /root/repo/gpurun_out/${TAG}_conv_relu_probe.jsonl 2> /root/repo/gpurun_out/${TAG}_conv_relu_probe.err )
            cp $(find /tmp/pcr_$TAG -name "*kernel_stats*" | head -1) gpurun_out/${TAG}_conv_relu_probe_kernel_stats.csv 2>/dev/null; cut -c1-400 gpurun_out/${TAG}_conv_relu_probe.jsonl ;;
    pmc_cold) ( cd /tmp && rm -rf /tmp/pmccw && mkdir -p /tmp/pmccw && for T in 1 0; do i=0; for CN in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do i=$((i+1)); timeout 200 rocprofv3 --pmc $CN --kernel-trace -d /tmp/pmccw/cold${T}_group$i -o p --output-format csv -- python /root/repo/tools/pmc_cold_warm.py run --cold $T > /tmp/pmccw_log_${T}_$i.txt 2>&1 || { echo "pass cold=$T group $i failed"; tail -3 /tmp/pmccw_log_${T}_$i.txt; }; done; done )
            python tools/pmc_cold_warm.py parse /tmp/pmccw > gpurun_out/${TAG}_pmc_cold_warm.jsonl; cat gpurun_out/${TAG}_pmc_cold_warm.jsonl ;;
    cpu_threads) timeout 900 python bench.py --cpu_threads_sweep 8,32,64,128,256 > gpurun_out/${TAG}_cpu_threads.txt 2> gpurun_out/${TAG}_cpu_threads.err; cut -c1-200 gpurun_out/${TAG}_cpu_threads.txt ;;
    dp)     timeout 900 python -m pytest tests/test_gpu_dp_product.py -m gpu -q -s --tb=short > gpurun_out/${TAG}_pytest_dp_product.log 2>&1; grep -E "world 2|passed|failed|Error|error" gpurun_out/${TAG}_pytest_dp_product.log | tail -12 ;;
    bench8) # `python bench.py --gpus 8` on ONE GPU: 8 gloo ranks (functional run of the N = 8 plumbing), wall time with the conv
            # find pass staggered (rank 0 first; default) and not, each from a cold MIOpen user db
            for SG in 1 0; do
              rm -rf ~/.config/miopen ~/.cache/miopen; t0=$(date +%s.%N)
              UH_FIND_STAGGER=$SG UH_DIST_BACKEND=gloo timeout 900 python bench.py --gpus ${A1:-8} --steps 5 --warmup 2 --cpu_baseline 0 --quality 0 > gpurun_out/${TAG}_bench_line_${A1:-8}ranks_gloo_one_gpu_stagger$SG.json 2> gpurun_out/${TAG}_bench_${A1:-8}ranks_stagger$SG.err
              echo "rc $? stagger $SG wall $(python -c "import time;print(round(time.time()-$t0,1))") s" | tee -a gpurun_out/${TAG}_bench_${A1:-8}ranks_wall.txt
              cut -c1-300 gpurun_out/${TAG}_bench_line_${A1:-8}ranks_gloo_one_gpu_stagger$SG.json; tail -3 gpurun_out/${TAG}_bench_${A1:-8}ranks_stagger$SG.err
            done ;;
    cold)   timeout 600 python tools/cold_forward.py --tag $TAG ${A1:+--batches ${A1//,/,}} > gpurun_out/${TAG}_cold_forward.jsonl 2> gpurun_out/${TAG}_cold_forward.err; cat gpurun_out/${TAG}_cold_forward.jsonl | cut -c1-260; tail -3 gpurun_out/${TAG}_cold_forward.err ;;
    train_ref) TAG=$TAG bash tools/train_reference_schedule.sh ;;
    dpsmooth) for ND in 0 1; do UH_TEST_TEXTURE=smooth UH_TEST_NONDET=$ND timeout 600 python -m pytest tests/test_gpu_dp_product.py -m gpu -q -s --tb=line -k "l1_equals or h_loss" > gpurun_out/${TAG}_pytest_dp_smooth_nondet$ND.log 2>&1; echo "smooth texture, default (non-deterministic) solvers allowed = $ND"; grep -E "world 2|passed|failed" gpurun_out/${TAG}_pytest_dp_smooth_nondet$ND.log | cut -c1-700; done ;;
    dpnoise) UH_TEST_NONDET=1 timeout 900 python -m pytest tests/test_gpu_dp_product.py -m gpu -q -s --tb=line > gpurun_out/${TAG}_pytest_dp_product_default_solvers.log 2>&1; grep -E "world 2|passed|failed" gpurun_out/${TAG}_pytest_dp_product_default_solvers.log | cut -c1-900 ;;
    power)  timeout 900 python tools/power_ab.py --libs ${A1:-shipped} --kernel ${A2:-bwd} --seconds 4 --reps 2 > gpurun_out/${TAG}_power_${A2:-bwd}.jsonl 2> gpurun_out/${TAG}_power.err; cat gpurun_out/${TAG}_power_${A2:-bwd}.jsonl ;;
    coldv)  : > gpurun_out/${TAG}_cold_forward_variants.jsonl
            for rep in 1 2; do
              timeout 300 python tools/cold_forward.py --tag shipped --batches 64,128 2>/dev/null | grep -v fit >> gpurun_out/${TAG}_cold_forward_variants.jsonl
              for V in ${A1//,/ }; do UH_LIB_PATH=$VDIR/libuh_$V.so timeout 300 python tools/cold_forward.py --tag $V --batches 64,128 2>/dev/null | grep -v fit >> gpurun_out/${TAG}_cold_forward_variants.jsonl; done
            done
            python - <<PYEOF
import json
for l in open('gpurun_out/${TAG}_cold_forward_variants.jsonl'):
    d = json.loads(l); print(d['tag'], d['B'], d['law'], d['temp'], d['us'], d['frac'])
PYEOF
            ;;
    *) echo "unknown stage $NAME" ;;
  esac
done
echo done
