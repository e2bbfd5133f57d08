#!/bin/bash
# round 3: Winograd patch stages in 16-byte LDS-DMA pieces (one per thread and chunk) against four dwords
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_wino.py -m gpu -x -q > gpurun_out/r3p16_pytest.log 2>&1 || { tail -40 gpurun_out/r3p16_pytest.log; exit 1; }
tail -2 gpurun_out/r3p16_pytest.log
OUT=gpurun_out/r3p16_patch.txt
: > $OUT
for m in dword 16 dword 16; do
  echo "== PCONV_WINO_PATCH=$m" >> $OUT
  PCONV_WINO_PATCH=$m PCONV_PROBE_SHORT=1 timeout -k 10 200 python tools/gpu_probe_wino.py >> $OUT 2>gpurun_out/r3p16_err.log || { tail -5 gpurun_out/r3p16_err.log; exit 1; }
done
cat $OUT
