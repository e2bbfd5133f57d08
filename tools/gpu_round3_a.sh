#!/bin/bash
# round 3, session A: decode(k+1) || synthesis(k) matrix -- DECODE_CHUNK 0/2/4 x frames 2/4/8, alternating on one box
set -e
mkdir -p gpurun_out
OUT=gpurun_out/r3a_overlap.txt
: > $OUT
python - <<'PY' >> gpurun_out/r3a_overlap.txt
import torch; print("device:", torch.cuda.get_device_name(0))
PY
for rep in 1 2; do
for f in 8 4 2; do
for c in 0 2 4; do
  if [ $c -ge $f ] && [ $c -ne 0 ]; then continue; fi
  echo "== rep $rep frames $f decode_chunk $c" >> $OUT
  PCONV_DECODE_CHUNK=$c timeout -k 10 300 python bench.py --steps 3 --frames-per-gpu $f --no-cpu-baseline 2>>gpurun_out/r3a_err.log \
    | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], 'MPix/s', d['ms_per_step'], 'ms/step', 'conv_s', d['config']['tile_conv_s_per_step'])" >> $OUT
  tail -1 $OUT
done
done
done
echo done
