#!/bin/bash
mkdir -p gpurun_out
OUT=gpurun_out/r3u_split.txt
: > $OUT
for cfg in "3 8" "2 9" "1 10" "6 5" "3 8" "1 10"; do
  set -- $cfg
  echo "== analysis_split $1 synthesis_split $2" >> $OUT
  PCONV_ANALYSIS_SPLIT=$1 PCONV_SYNTHESIS_SPLIT=$2 timeout -k 10 300 python bench.py --steps 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], 'MPix/s', d['ms_per_step'], 'ms/step', 'conv_s', d['config']['tile_conv_s_per_step'])" >> $OUT
done
cat $OUT
