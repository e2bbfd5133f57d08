#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "streamed_1x1" > $O/r4w_tests.txt 2>&1; rc=$?
tail -5 $O/r4w_tests.txt
[ $rc = 0 ] || exit 1
for tn in 128; do for rep in 1 2; do for v in tiled quads; do
  export PCONV_CONV1X1=tiled PROBE_TN=$tn; unset PCONV_CONV1X1_WAYOUT
  [ $v = quads ] && export PCONV_CONV1X1_WAYOUT=quads
  echo "== $v tn $tn rep $rep"; timeout -k 10 120 python tools/gpu_probe_1x1.py 2>&1 | grep -v "Warning\|amdgpu.ids"
done; done; done 2>&1 | tee $O/r4w_1x1_quads.txt
