#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "streamed_1x1" > $O/r4s_tests.txt 2>&1; rc=$?
tail -5 $O/r4s_tests.txt
[ $rc = 0 ] || exit 1
export PROBE_TN=128
for rep in 1 2; do for v in tiled stream sd2; do
  unset PCONV_HIP_LIB; export PCONV_CONV1X1=stream
  case $v in tiled) export PCONV_CONV1X1=tiled;; stream) ;; *) export PCONV_HIP_LIB=$PWD/tools/_build/libpconv_hip_$v.so;; esac
  echo "== $v rep $rep"; timeout -k 10 120 python tools/gpu_probe_1x1.py 2>&1 | grep -v "Warning\|amdgpu.ids\|768"
done; done 2>&1 | tee $O/r4v_1x1_depth.txt
unset PROBE_TN
PCONV_HIP_LIB=$PWD/tools/_build/libpconv_hip_sstamp.so PCONV_CONV1X1=stream timeout -k 10 200 python tools/gpu_probe_stream_stamps.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/r4v_stream_stamps.txt
