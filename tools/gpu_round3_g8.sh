#!/bin/bash
# round 3: decoder groups beyond four (8 frames per GPU)
set -e
mkdir -p gpurun_out
OUT=gpurun_out/r3g8_groups.txt
: > $OUT
for g in 4 6 8 4 8; do
  echo "== PCONV_ENGINE_GROUPS=$g PCONV_ENGINE_CHAIN=host" >> $OUT
  PCONV_ENGINE_GROUPS=$g PCONV_ENGINE_CHAIN=host PCONV_ENGINE_TRACE=1 timeout -k 10 300 python bench.py --steps 3 --no-cpu-baseline 2>gpurun_out/r3g8_err.log \
    | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], 'MPix/s', d['ms_per_step'], 'ms/step')" >> $OUT
  grep "decode 8 frame" gpurun_out/r3g8_err.log | tail -1 >> $OUT || true
done
cat $OUT
