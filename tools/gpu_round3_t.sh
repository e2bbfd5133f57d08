#!/bin/bash
mkdir -p gpurun_out
OUT=gpurun_out/r3t_knobs.txt
: > $OUT
for cfg in "PCONV_EE_PPW=8" "PCONV_EE_PPW=4" "PCONV_EE_PPW=2" "PCONV_EE_PPW=16" "PCONV_EE_JOINT=1" "PCONV_ENGINE_SPIN_US=200" "PCONV_ENCODE_CHUNK=4" "PCONV_EE_PPW=8"; do
  echo "== $cfg" >> $OUT
  env $cfg PCONV_ENGINE_TIMING=1 timeout -k 10 300 python bench.py --steps 3 --no-cpu-baseline 2>gpurun_out/r3t_err.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], 'MPix/s', d['ms_per_step'], 'ms/step')" >> $OUT
  grep "pconv engine\] decode" gpurun_out/r3t_err.log | tail -2 | head -1 | cut -c1-120 >> $OUT
done
cat $OUT
