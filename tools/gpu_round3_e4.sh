#!/bin/bash
# round 3: masked gather offsets in the step kernel, A/B on one box
set -e
mkdir -p gpurun_out
OUT=gpurun_out/r3e4_masked_taps.txt
: > $OUT
for m in 0 1 0 1; do
  echo "== PCONV_EE_MASKED_TAPS=$m" >> $OUT
  PCONV_EE_MASKED_TAPS=$m timeout -k 10 300 python tools/gpu_probe_engine.py --batch --batch8 2>gpurun_out/r3e4_err.log | grep "2048x4096" | grep rep1 | cut -c1-110 >> $OUT
done
cat $OUT
