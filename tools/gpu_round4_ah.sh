#!/bin/bash
O=$PWD/gpurun_out
for rep in 1 2; do for x in 0 1; do
  echo "== PCONV_WINO_XCD=$x rep $rep"
  PCONV_WINO_XCD=$x PCONV_PROBE_NODIRECT=1 timeout -k 10 200 python tools/gpu_probe_wino42.py 2>&1 | grep "3x3" | sed 's/wino [0-9.]* ms ([0-9]* TF alg, diff 0)//'
done; done 2>&1 | tee $O/r4ah_wino42_xcd.txt
for x in 0 1; do PCONV_WINO_XCD=$x python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xcd $x', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; done | tee -a $O/r4ah_wino42_xcd.txt
