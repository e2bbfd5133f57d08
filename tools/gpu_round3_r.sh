#!/bin/bash
mkdir -p gpurun_out
OUT=gpurun_out/r3r_groups.txt
: > $OUT
for rep in 1 2; do
for cfg in "PCONV_ENGINE_GROUPS=2 PCONV_ENGINE_CHAIN=host" "PCONV_ENGINE_GROUPS=3 PCONV_ENGINE_CHAIN=host" "PCONV_ENGINE_GROUPS=4 PCONV_ENGINE_CHAIN=host"; do
  echo "== rep $rep $cfg" >> $OUT
  env $cfg PCONV_ENGINE_TIMING=1 timeout -k 10 300 python bench.py --steps 3 --no-cpu-baseline 2>gpurun_out/r3r_err.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], 'MPix/s', d['ms_per_step'], 'ms/step')" >> $OUT
  grep "pconv engine\] decode" gpurun_out/r3r_err.log | tail -2 | head -1 >> $OUT
done
done
for f in 4 2; do
for cfg in "PCONV_ENGINE_GROUPS=2" "PCONV_ENGINE_GROUPS=3 PCONV_ENGINE_CHAIN=host" "PCONV_ENGINE_GROUPS=4 PCONV_ENGINE_CHAIN=host"; do
  echo "== frames $f $cfg" >> $OUT
  env $cfg timeout -k 10 300 python bench.py --steps 3 --no-cpu-baseline --frames-per-gpu $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], 'MPix/s', d['ms_per_step'], 'ms/step')" >> $OUT
done
done
cat $OUT
