#!/bin/bash
O=$PWD/gpurun_out
for rep in 1 2 3; do for r in 1 4; do
  PCONV_ENGINE_ENCODE_RANGES=$r timeout -k 10 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ranges $r rep $rep:', d['value'], d['ms_per_step'])"
done; done 2>&1 | tee $O/r4am_encode_ranges_ab.txt
