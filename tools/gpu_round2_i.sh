#!/bin/bash
# session I: step kernel geometry sweep (threads per workgroup x positions per wave)
set -e
mkdir -p gpurun_out
for cfg in "256 4" "256 2" "512 2" "512 1" "1024 1" "1024 2" "512 3"; do
  set -- $cfg
  echo "== block $1 ppw $2"
  PCONV_EE_BLOCK=$1 PCONV_EE_PPW=$2 timeout -k 10 200 python tools/gpu_probe_engine.py --batch --batch8 2>&1 | grep "rep1" | sed 's/bytes.*//'
done
echo done
