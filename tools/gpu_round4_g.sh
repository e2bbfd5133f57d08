#!/bin/bash
# round 4, call G: Winograd F(4x2,3x3) -- parity tests, then timing against F(2x2,3x3)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_wino42.py -m gpu -x -q > gpurun_out/r4g_tests.log 2>&1 || { tail -50 gpurun_out/r4g_tests.log; exit 1; }
tail -3 gpurun_out/r4g_tests.log
PCONV_PROBE_NODIRECT=1 timeout -k 10 300 python tools/gpu_probe_wino42.py 2>&1 | tee gpurun_out/r4g_wino42.txt
