#!/bin/bash
# round 3, session W4: resident 1x1 kernel with its second wave per SIMD held back (way out of one wave
# beside the matrix loop of its partner)
set -e
mkdir -p gpurun_out
OUT=gpurun_out/r3w4_stagger.txt
: > $OUT
for st in tiled 0 2 4 8 16; do
  echo "== stagger $st" >> $OUT
  if [ $st = tiled ]; then
    PCONV_CONV1X1=tiled timeout -k 10 200 python tools/gpu_probe_1x1.py >> $OUT 2>gpurun_out/r3w4_err.log
  else
    PCONV_CONV1X1=resident PCONV_CONV1X1_STAGGER=$st timeout -k 10 200 python tools/gpu_probe_1x1.py >> $OUT 2>gpurun_out/r3w4_err.log
  fi
done
cat $OUT
