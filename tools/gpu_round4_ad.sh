#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_wino42.py -x -q -m gpu > $O/r4ad_tests.txt 2>&1; rc=$?
tail -3 $O/r4ad_tests.txt
[ $rc = 0 ] || exit 1
for rep in 1 2; do PCONV_PROBE_SHORT=1 PCONV_PROBE_NODIRECT=1 timeout -k 10 120 python tools/gpu_probe_wino42.py 2>&1 | grep "3x3"; done | tee $O/r4ad_wino42_pipelined_wayout.txt
