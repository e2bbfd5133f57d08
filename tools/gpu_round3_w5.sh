#!/bin/bash
# round 3, session W5: pipelined way out of the tiled 1x1 / GDN kernel: parity tests, old vs new
set -e
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -m gpu -x -q > gpurun_out/r3w5_pytest.log 2>&1 || { tail -40 gpurun_out/r3w5_pytest.log; exit 1; }
tail -2 gpurun_out/r3w5_pytest.log
OUT=gpurun_out/r3w5_1x1.txt
: > $OUT
for mode in batch pipe batch pipe; do
  echo "== way out: $mode" >> $OUT
  PCONV_CONV1X1_WAYOUT=$mode timeout -k 10 200 python tools/gpu_probe_1x1.py >> $OUT 2>gpurun_out/r3w5_err.log || { tail -5 gpurun_out/r3w5_err.log; exit 1; }
done
cat $OUT
