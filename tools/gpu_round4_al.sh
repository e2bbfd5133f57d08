#!/bin/bash
O=$PWD/gpurun_out
for r in 1 4; do
  echo "== 1 frame, ranges $r"
  PCONV_ENGINE_ENCODE_RANGES=$r PCONV_ENGINE_TIMING=1 timeout -k 10 200 python bench.py --frames-per-gpu 1 --steps 3 --warmup 1 --no-cpu-baseline --no-check 2> $O/r4al_err_$r.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  grep "pconv engine" $O/r4al_err_$r.txt | tail -4 | cut -c1-150
done 2>&1 | tee $O/r4al_one_frame_ranges.txt
