#!/bin/bash
# streamed 1x1 kernel: parity against the tiled kernel, then timing at the codec's shapes
set -o pipefail
O=$PWD/gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "streamed_1x1" > $O/r4s_tests.txt 2>&1; rc=$?
tail -15 $O/r4s_tests.txt
[ $rc = 0 ] || exit 1
for tn in 16 128; do for rep in 1 2; do for v in tiled stream; do
  unset PCONV_HIP_LIB; export PCONV_CONV1X1=$v PROBE_TN=$tn
  echo "== $v tn $tn rep $rep"; timeout -k 10 120 python tools/gpu_probe_1x1.py 2>&1 | grep -v "Warning\|amdgpu.ids\|768"
done; done; done 2>&1 | tee $O/r4s_1x1_stream.txt
unset PROBE_TN
tools/gpu_round4_u.sh
