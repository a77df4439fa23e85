#!/bin/bash
# Functional end-to-end demo: train on fresh in-HBM synthetic pairs, then the reference's test statistics.
cd /root/repo; mkdir -p gpurun_out
M=unsuperviseddeephomographyral2018_amd.homography_CNN_synthetic
for LT in h_loss l1_loss; do
  echo "=== $LT ==="
  timeout 300 python -m $M --mode train --loss_type $LT --batch_size 64 --num_total_steps ${STEPS:-2500} --log_every 500 \
      --save_every 1000000 --model_dir /tmp/uh_models --fused_patch True 2>/dev/null | grep -E "Train:|Decay"
  timeout 120 python -m $M --mode test --loss_type $LT --batch_size 64 --num_test_data 512 --model_dir /tmp/uh_models 2>/dev/null | grep -E "Result|Average"
done > gpurun_out/train_demo.txt 2>&1
