#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "small_cout" > $O/r4z_tests.txt 2>&1; rc=$?
tail -8 $O/r4z_tests.txt
[ $rc = 0 ] || exit 1
for rep in 1 2; do for v in 0 1; do PCONV_CONV_SMALL=$v timeout -k 10 120 python tools/gpu_probe_small.py 2>&1 | grep "3x3"; done; done | tee $O/r4z_small_cout.txt
