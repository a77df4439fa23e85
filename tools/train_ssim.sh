#!/bin/bash
# ssim_loss diverged at lr 3e-4 (profiles/r02_train_all_losses.txt): the reference's default lr 1e-4 and a smaller one
cd /root/repo; mkdir -p gpurun_out
M=unsuperviseddeephomographyral2018_amd.homography_CNN_synthetic
OUT=gpurun_out/r02_train_ssim.txt
echo "# ssim_loss, 8 000 steps, B=64, multi-octave textures; identity error = 26 px" > $OUT
for LR in "1e-4 9e-5" "3e-5 2.7e-5"; do
  set -- $LR
  echo "=== --loss_type ssim_loss --lr $1" >> $OUT
  timeout 300 python -m $M --mode train --loss_type ssim_loss --batch_size 64 --num_total_steps 8000 --log_every 2000 \
      --save_every 100000000 --model_dir /tmp/uh_models_ssim_$1 --data_pool 256 --texture multiscale --lr $1 --min_lr $2 2>&1 \
      | grep -E "Train:|rror" | sed 's/rec_loss.*lr/lr/' >> $OUT
  timeout 200 python -m $M --mode test --save_visual False --loss_type ssim_loss --batch_size 64 --num_test_data 1024 --model_dir /tmp/uh_models_ssim_$1 --texture multiscale 2>&1 \
      | grep -E "Result|Average|rror" >> $OUT
done
cat $OUT
