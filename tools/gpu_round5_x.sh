#!/bin/bash
# Round 5, X: the transforms' batched / frame-wise split points again (round 3 chose 3 / 8 with the direct kernels)
O=$PWD/gpurun_out/r5x
mkdir -p $O
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'MPix/s', d['value'], 'ms/step', d['ms_per_step'])"; }
for cfg in "3 8" "2 8" "1 8" "3 9" "3 10" "3 6" "5 8" "3 8"; do
  set -- $cfg
  PCONV_ANALYSIS_SPLIT=$1 PCONV_SYNTHESIS_SPLIT=$2 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check 2> $O/err.txt | line "analysis split $1 synthesis split $2:" | tee -a $O/splits.txt || tail -3 $O/err.txt
done
exit 0
