#!/bin/bash
# round 3, session K: engine knobs with the faster transforms (encode chunk, decode groups / chain)
mkdir -p gpurun_out
OUT=gpurun_out/r3k_knobs.txt
: > $OUT
for cfg in "default" "PCONV_ENCODE_CHUNK=4" "PCONV_ENCODE_CHUNK=8" "PCONV_ENGINE_GROUPS=4" "PCONV_ENGINE_CHAIN=queued" "PCONV_ENGINE_GROUPS=4 PCONV_ENGINE_CHAIN=queued" "default"; do
  echo "== $cfg" >> $OUT
  if [ "$cfg" = default ]; then
    PCONV_ENGINE_TIMING=1 timeout -k 10 300 python bench.py --steps 3 --no-cpu-baseline 2>gpurun_out/r3k_err.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], 'MPix/s', d['ms_per_step'], 'ms/step')" >> $OUT
  else
    env $cfg PCONV_ENGINE_TIMING=1 timeout -k 10 300 python bench.py --steps 3 --no-cpu-baseline 2>gpurun_out/r3k_err.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], 'MPix/s', d['ms_per_step'], 'ms/step')" >> $OUT
  fi
  grep "pconv engine" gpurun_out/r3k_err.log | tail -5 >> $OUT
done
cat $OUT
