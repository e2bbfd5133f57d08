#!/bin/bash
# round 3, session W2: the weight-resident 1x1 kernel with the pipelined way out against the tiled kernel
set -e
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -m gpu -x -q > gpurun_out/r3w2_pytest.log 2>&1 || { tail -40 gpurun_out/r3w2_pytest.log; exit 1; }
tail -2 gpurun_out/r3w2_pytest.log
OUT=gpurun_out/r3w2_1x1.txt
: > $OUT
for mode in tiled resident tiled resident; do
  echo "== $mode" >> $OUT
  PCONV_CONV1X1=$mode timeout -k 10 200 python tools/gpu_probe_1x1.py >> $OUT 2>gpurun_out/r3w2_err.log || { tail -5 gpurun_out/r3w2_err.log; exit 1; }
done
cat $OUT
