#!/bin/bash
O=$PWD/gpurun_out
for n in 1 2; do for r in 1 2 4; do
  echo "== N=$n ranges $r"
  PCONV_ENGINE_ENCODE_RANGES=$r PCONV_ENGINE_TIMING=1 timeout -k 10 200 python tools/gpu_probe_entropy_only.py $n 4 encode 2>&1 | grep "entropy encode\|pconv engine" | tail -4 | cut -c1-140
done; done 2>&1 | tee $O/r4ak_encode_ranges_probe.txt
