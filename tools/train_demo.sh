#!/bin/bash
# Functional end-to-end demo (round 2): does the loop learn homographies?  Unsupervised photometric L1 (the hot path's
# gradients) from scratch on multi-octave synthetic textures cycled from an in-HBM pool, next to the supervised 4-pt loss;
# then the reference's test statistics (mean corner error, failure rate) on held-out synthetic pairs.
cd /root/repo; mkdir -p gpurun_out
M=unsuperviseddeephomographyral2018_amd.homography_CNN_synthetic
OUT=gpurun_out/${TAG:-r02}_train_demo.txt
: > $OUT
run() {  # name, loss, steps, extra
  echo "=== $1 : --loss_type $2 --num_total_steps $3 $4" >> $OUT
  timeout 900 python -m $M --mode train --loss_type $2 --batch_size 64 --num_total_steps $3 --log_every ${LOG:-2500} \
      --save_every 100000000 --model_dir /tmp/uh_models_$1 --data_pool ${POOL:-512} --texture multiscale $4 2>&1 \
      | grep -E "Train:|rror" | sed 's/rec_loss.*lr/lr/' >> $OUT
  timeout 200 python -m $M --mode test --save_visual False --loss_type $2 --batch_size 64 --num_test_data 1024 --model_dir /tmp/uh_models_$1 --texture multiscale 2>&1 \
      | grep -E "Result|Average|ercentile|rror" >> $OUT
}
run unsup_lr1e-4 l1_loss ${STEPS:-30000} ""
run unsup_lr3e-4 l1_loss ${STEPS:-30000} "--lr 3e-4 --min_lr 2.7e-4"
run sup_lr1e-4 h_loss 8000 ""
echo done
