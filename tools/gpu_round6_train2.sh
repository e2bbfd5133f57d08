#!/bin/bash
# Round 6, training call 2: recover stage 1 from the first call's (spiked) state at a lower learning rate, keep the
# BEST epoch, then stage 2 (entropy model on the frozen codes), export, and the evaluation through the codec.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/train_r6
mkdir -p $O
cd $R
python tools/train_round6.py stage1 --minutes ${S1_MIN:-5.5} --lr ${S1_LR:-5e-5} --clip 0.5 --workers 4 --from trained/r6/stage1_a.pack.pt > $O/stage1b_stdout.txt 2>&1 || { tail -20 $O/stage1b_stdout.txt; exit 1; }
grep "Test set" $O/stage1b_stdout.txt | tail -8
tail -1 $O/stage1b_stdout.txt
python tools/train_round6.py stage2 --minutes ${S2_MIN:-7} --lr 1e-4 --clip 0.5 --workers 4 --from $O/stage1.pack.pt > $O/stage2_stdout.txt 2>&1 || { tail -20 $O/stage2_stdout.txt; exit 1; }
grep "Test set" $O/stage2_stdout.txt | tail -8
tail -2 $O/stage2_stdout.txt
cat $O/stage2_report.json
