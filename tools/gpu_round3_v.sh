#!/bin/bash
# round 3, session V: Winograd kernel variants against the committed one (tools/_build/libpconv_hip_v8.so), same box
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_wino.py -m gpu -x -q > gpurun_out/r3v_pytest.log 2>&1 || { tail -40 gpurun_out/r3v_pytest.log; exit 1; }
tail -2 gpurun_out/r3v_pytest.log
OUT=gpurun_out/r3v_variants.txt
: > $OUT
for lib in v8 new v8 new $EXTRA_VARIANTS; do
  echo "== $lib" >> $OUT
  if [ $lib = new ]; then unset PCONV_HIP_LIB; else export PCONV_HIP_LIB=$PWD/tools/_build/libpconv_hip_$lib.so; fi
  PCONV_PROBE_SHORT=1 timeout -k 10 200 python tools/gpu_probe_wino.py >> $OUT 2>gpurun_out/r3v_err.log || { tail -5 gpurun_out/r3v_err.log; exit 1; }
done
cat $OUT
