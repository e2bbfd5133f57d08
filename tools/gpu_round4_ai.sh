#!/bin/bash
O=$PWD/gpurun_out
for rep in 1 2; do for v in default nointr; do
  unset HSA_ENABLE_INTERRUPT
  [ $v = nointr ] && export HSA_ENABLE_INTERRUPT=0
  echo "== $v rep $rep"
  PCONV_ENGINE_TIMING=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check 2> $O/r4ai_err_$v.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  grep "decode 8" $O/r4ai_err_$v.txt | tail -2
done; done 2>&1 | tee $O/r4ai_hsa_interrupt.txt
