#!/bin/bash
# Round 5: decoder groups at 8 frames per GPU again (final tree): PCONV_ENGINE_GROUPS x PCONV_ENGINE_CHAIN
O=$PWD/gpurun_out/r5g
mkdir -p $O
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'MPix/s', d['value'], 'ms/step', d['ms_per_step'])"; }
for cfg in "4 host" "2 host" "2 queued" "3 host" "4 queued" "4 host"; do
  set -- $cfg
  PCONV_ENGINE_GROUPS=$1 PCONV_ENGINE_CHAIN=$2 PCONV_ENGINE_TIMING=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check 2> $O/err.txt | line "groups $1 chain $2:" | tee -a $O/groups.txt
  grep "decode 8" $O/err.txt | tail -1 | cut -c1-170 | tee -a $O/groups.txt
done
exit 0
