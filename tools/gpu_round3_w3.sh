#!/bin/bash
# round 3, session W3: ablations of the tiled 1x1 / GDN kernel (timing only)
set -e
mkdir -p gpurun_out
OUT=gpurun_out/r3w3_1x1_ablation.txt
: > $OUT
for lib in ${VARIANTS:-new cabl_NOEPI cabl_NOXDMA cabl_NOWDMA new}; do
  echo "== $lib" >> $OUT
  if [ $lib = new ]; then unset PCONV_HIP_LIB; else export PCONV_HIP_LIB=$PWD/tools/_build/libpconv_hip_$lib.so; fi
  timeout -k 10 200 python tools/gpu_probe_1x1.py >> $OUT 2>gpurun_out/r3w3_err.log || { tail -5 gpurun_out/r3w3_err.log; exit 1; }
done
cat $OUT
