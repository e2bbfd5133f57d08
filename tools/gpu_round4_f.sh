#!/bin/bash
# round 4, call F: decoder knobs with the band step kernel (groups / chain / rows per workgroup), N = 8 and 4
set -o pipefail
mkdir -p gpurun_out
for rows in 8 16; do
for groups in 1 2 4 8; do
for chain in queued host; do
  echo "rows $rows groups $groups chain $chain: $(PCONV_EE_ROWS=$rows PCONV_ENGINE_GROUPS=$groups PCONV_ENGINE_CHAIN=$chain timeout -k 10 120 python tools/gpu_probe_entropy_only.py 8 2 2>&1 | tail -1)"
done; done; done 2>&1 | tee gpurun_out/r4f_knobs.txt
