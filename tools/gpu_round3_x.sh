#!/bin/bash
# round 3, session X: ablations of the current Winograd kernel (timing only: results are wrong by construction)
set -e
mkdir -p gpurun_out
OUT=gpurun_out/r3x_ablation.txt
: > $OUT
for lib in ${VARIANTS:-new abl_NOEPILOGUE abl_NOTRANSFORM abl_NOPDMA abl_NOBAR abl_NOEPI_NOTR new}; do
  echo "== $lib" >> $OUT
  if [ $lib = new ]; then unset PCONV_HIP_LIB; else export PCONV_HIP_LIB=$PWD/tools/_build/libpconv_hip_$lib.so; fi
  PCONV_PROBE_SHORT=1 timeout -k 10 200 python tools/gpu_probe_wino.py >> $OUT 2>gpurun_out/r3x_err.log || { tail -5 gpurun_out/r3x_err.log; exit 1; }
done
cat $OUT
