#!/bin/bash
# ONE training run at the reference's OWN schedule (homography_CNN_synthetic.py:79-81,159-169: 150 000 steps, Adam lr 1e-4,
# exponential_decay 0.96 every 58 117 steps staircase, dropout 0.5) on the unsupervised photometric l1_loss, batch 64,
# 240x320 / 128x128 patch / RHO 45, from scratch, then the reference's test loop (:391-401,573-579) on held-out pairs.
# Data: in-HBM synthetic pairs (multiscale texture, sampling law of gen_synthetic_data.py:42-64), pool of $POOL batches
# cycled in a fresh random order per pass.  ~14 GPU-minutes at 5.5 ms / step.
cd /root/repo; mkdir -p gpurun_out
M=unsuperviseddeephomographyral2018_amd.homography_CNN_synthetic
TAG=${TAG:-r04}; STEPS=${STEPS:-150000}; POOL=${POOL:-1024}; BATCH=${BATCH:-64}; LOSS=${LOSS:-l1_loss}   # LOSS=h_loss: the SUPERVISED 4-pt regression (reference flag --loss_type h_loss)
OUT=gpurun_out/${TAG}_train_reference_schedule$([ "$LOSS" = l1_loss ] || echo _$LOSS).txt
MD=/tmp/uh_models_refsched
echo "=== reference schedule: --loss_type $LOSS --lr 1e-4 --min_lr 0.9e-4 --batch_size $BATCH --num_total_steps $STEPS --data_pool $POOL --texture multiscale (from scratch)" > $OUT
t0=$(date +%s)
timeout ${TRAIN_TIMEOUT:-1500} python -m $M --mode train --loss_type $LOSS --batch_size $BATCH --num_total_steps $STEPS --log_every 10000 \
    --save_every 100000000 --model_dir $MD --data_pool $POOL --texture multiscale --lr 1e-4 --min_lr .9e-4 2>&1 \
    | grep -E "Train:|rror|Decay" | sed 's/rec_loss.*lr/lr/' >> $OUT
echo "=== training wall time: $(( $(date +%s) - t0 )) s" >> $OUT
timeout 300 python -m $M --mode test --save_visual False --loss_type $LOSS --batch_size $BATCH --num_test_data 1024 --model_dir $MD --texture multiscale 2>&1 \
    | grep -E "Result|Average|ercentile|rror" >> $OUT
tail -8 $OUT
