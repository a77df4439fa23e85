#!/bin/bash
# 100 000 unsupervised steps (photometric L1, from scratch) on a 1024-batch in-HBM pool (120 GB): how far does the loop get?
cd /root/repo; mkdir -p gpurun_out
M=unsuperviseddeephomographyral2018_amd.homography_CNN_synthetic
OUT=gpurun_out/${TAG:-r02}_train_long.txt
echo "=== unsup 100k : --loss_type l1_loss --lr 3e-4 --min_lr 2.7e-4 --batch_size 64 --data_pool 1024 --texture multiscale" > $OUT
timeout 1500 python -m $M --mode train --loss_type l1_loss --batch_size 64 --num_total_steps 100000 --log_every 10000 \
    --save_every 100000000 --model_dir /tmp/uh_models_long --data_pool 1024 --texture multiscale --lr 3e-4 --min_lr 2.7e-4 2>&1 \
    | grep -E "Train:|rror|Decay" | sed 's/rec_loss.*lr/lr/' >> $OUT
timeout 300 python -m $M --mode test --save_visual False --loss_type l1_loss --batch_size 64 --num_test_data 2048 --model_dir /tmp/uh_models_long --texture multiscale 2>&1 \
    | grep -E "Result|Average|ercentile|rror" >> $OUT
echo done
