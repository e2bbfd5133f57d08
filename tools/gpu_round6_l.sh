#!/bin/bash
# Round 6, L: (1) the trained model through the per-op loops == the engine's file (new test); (2) the DDP training path
# with TWO ranks on the one GPU (gloo carries the gradient all-reduce of GPU tensors: RCCL refuses two ranks per
# device) -- the Job of train.py end to end on the HIP ops with gradient synchronisation, a rehearsal of f4's N > 1 path.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6l
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_trained.py -x -q -m gpu 2>&1 | tail -3
rm -rf /tmp/ddp2
GPU_MAX_HW_QUEUES=8 timeout -k 10 500 python -m pseudocylindrical_convolution_amd.train --gpus 2 --share-gpu --dist-backend gloo \
  --base --procedural 32 --height 512 --width 1024 --batch-size 1 --test-batch-size 1 --acc-batch 1 --valid-dim 56 --epochs 100000 \
  --time-budget 60 --lr 1e-4 --beta 0.01 --clip 1 --mean 0 --workers 0 --base-dir /tmp/ddp2 --seed 6 --verbose > $O/ddp2_stdout.txt 2>&1
tail -4 $O/ddp2_stdout.txt | cut -c1-200
cat /tmp/ddp2/save_models/base_opt_192_56_16_logs_0.txt | grep -E "Test set|checksum|time budget" | tee $O/ddp2_log.txt
grep -c "Train Epoch" /tmp/ddp2/save_models/base_opt_192_56_16_logs_0.txt
echo done
