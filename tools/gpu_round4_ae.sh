#!/bin/bash
O=$PWD/gpurun_out
for rep in 1 2; do for us in 60 1000; do
  echo "== spin $us us rep $rep"
  PCONV_ENGINE_SPIN_US=$us PCONV_ENGINE_TIMING=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check 2> $O/r4ae_err_$us.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  grep "decode 8" $O/r4ae_err_$us.txt | tail -2
done; done 2>&1 | tee $O/r4ae_spin.txt
