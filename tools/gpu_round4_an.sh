#!/bin/bash
O=$PWD/gpurun_out
for rep in 1 2; do for cfg in "0 256" "4 256" "16 256" "0 512"; do
  set -- $cfg
  unset PCONV_EE_PPW PCONV_EE_BLOCK
  [ $1 != 0 ] && export PCONV_EE_PPW=$1
  export PCONV_EE_BLOCK=$2
  PCONV_ENGINE_TIMING=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check 2> $O/r4an_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ppw $1 block $2 rep $rep:', d['value'], d['ms_per_step'])"
  grep "decode 8" $O/r4an_err.txt | tail -1 | cut -c1-120
done; done 2>&1 | tee $O/r4an_decode_knobs.txt
