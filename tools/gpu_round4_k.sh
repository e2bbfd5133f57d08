#!/bin/bash
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_wino42.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do for v in w42a w42d; do
  echo "== $v rep $rep"; PCONV_HIP_LIB=$PWD/tools/_build/libpconv_hip_$v.so PCONV_PROBE_SHORT=1 PCONV_PROBE_NODIRECT=1 timeout -k 10 120 python tools/gpu_probe_wino42.py 2>&1 | grep "3x3" | sed 's/wino [0-9.]* ms ([0-9]* TF alg, diff 0)//'
done; done 2>&1 | tee gpurun_out/r4k_ab.txt
