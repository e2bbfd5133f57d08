#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_codec_vs_oracle.py -x -q -m gpu > $O/r4aj_tests.txt 2>&1; rc=$?
tail -4 $O/r4aj_tests.txt
[ $rc = 0 ] || exit 1
for rep in 1 2; do for r in 1 4 6; do
  echo "== PCONV_ENGINE_ENCODE_RANGES=$r rep $rep"
  PCONV_ENGINE_ENCODE_RANGES=$r PCONV_ENGINE_TIMING=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check 2> $O/r4aj_err_$r.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  grep "encode 2" $O/r4aj_err_$r.txt | tail -4 | cut -c1-150
done; done 2>&1 | tee $O/r4aj_encode_ranges.txt
for r in 1 4; do PCONV_ENGINE_ENCODE_RANGES=$r python bench.py --frames-per-gpu 1 --steps 3 --warmup 1 --no-cpu-baseline --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1 frame, ranges $r:', d['value'], d['ms_per_step'])"; done | tee -a $O/r4aj_encode_ranges.txt
