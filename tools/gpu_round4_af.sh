#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out
PCONV_ENGINE_POLL=1 timeout -k 10 600 python -m pytest tests/test_gpu_engine.py -x -q -m gpu > $O/r4af_tests.txt 2>&1; rc=$?
tail -3 $O/r4af_tests.txt
[ $rc = 0 ] || exit 1
for rep in 1 2; do for poll in 0 1; do
  echo "== poll $poll rep $rep"
  PCONV_ENGINE_POLL=$poll PCONV_ENGINE_TIMING=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check 2> $O/r4af_err_$poll.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  grep "decode 8" $O/r4af_err_$poll.txt | tail -2
done; done 2>&1 | tee $O/r4af_poll.txt
