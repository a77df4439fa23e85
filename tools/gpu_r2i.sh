#!/bin/bash
# Round-2 GPU session I: does the unsupervised (photometric L1) path learn homographies?  Short exploratory runs.
mkdir -p gpurun_out; cd /root/repo
M=unsuperviseddeephomographyral2018_amd.homography_CNN_synthetic
: > gpurun_out/r2i_explore.txt
run() {  # name, extra args
  echo "=== $1 : $2" >> gpurun_out/r2i_explore.txt
  timeout 400 python -m $M --mode train --loss_type l1_loss --batch_size 64 --num_total_steps ${STEPS:-8000} --log_every 1000 \
      --save_every 100000000 --model_dir /tmp/uh_models_$1 --data_pool 256 $2 2>&1 | grep -E "Train:|Error|error" | sed 's/rec_loss.*lr/lr/' >> gpurun_out/r2i_explore.txt
  timeout 120 python -m $M --mode test --loss_type l1_loss --batch_size 64 --num_test_data 512 --model_dir /tmp/uh_models_$1 $3 2>&1 | grep -E "Result|Average|rror" >> gpurun_out/r2i_explore.txt
}
run smooth "--texture smooth" "--texture smooth"
run multi "--texture multiscale" "--texture multiscale"
run multi_lr3 "--texture multiscale --lr 3e-4 --min_lr 2.7e-4" "--texture multiscale"
run multi_fused "--texture multiscale --fused_patch True" "--texture multiscale"
echo done
