#!/bin/bash
# Every photometric loss trains through its HIP gradient: 10 000 unsupervised steps from scratch per loss
# type on the default (full-frame) path, then the reference's test statistics on held-out pairs.
cd /root/repo; mkdir -p gpurun_out
M=unsuperviseddeephomographyral2018_amd.homography_CNN_synthetic
OUT=gpurun_out/r02_train_all_losses.txt
echo "# 10 000 steps each, B=64, lr 3e-4, multi-octave textures from a 512-batch in-HBM pool; identity error = 26 px" > $OUT
for L in rec_loss ssim_loss l1_smooth_loss ncc_loss l1_loss; do
  echo "=== --loss_type $L" >> $OUT
  timeout 400 python -m $M --mode train --loss_type $L --batch_size 64 --num_total_steps 10000 --log_every 5000 \
      --save_every 100000000 --model_dir /tmp/uh_models_$L --data_pool 512 --texture multiscale --lr 3e-4 --min_lr 2.7e-4 2>&1 \
      | grep -E "Train:|rror" | sed 's/rec_loss.*lr/lr/' >> $OUT
  timeout 200 python -m $M --mode test --save_visual False --loss_type $L --batch_size 64 --num_test_data 1024 --model_dir /tmp/uh_models_$L --texture multiscale 2>&1 \
      | grep -E "Result|Average|rror" >> $OUT
done
cat $OUT
