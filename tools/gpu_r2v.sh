#!/bin/bash
# session V: the non-finite-gradient guard test + loss-backward parity (float4 kernel), then the 100k-step unsupervised rerun
cd /root/repo; mkdir -p gpurun_out
timeout 600 python -m pytest tests -q -m gpu -x --tb=short -k "degenerate or patch_loss_backward or call_surface or graph_tail or hipgraph" > gpurun_out/r2v_tests.txt 2>&1
tail -5 gpurun_out/r2v_tests.txt
bash tools/train_long.sh
cat gpurun_out/r02_train_long.txt
