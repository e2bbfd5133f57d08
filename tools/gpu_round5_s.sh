#!/bin/bash
# Round 5, S: stamps of the four-block encoder kernel's workgroups (tools/gpu_probe_ee4_stamps.py)
O=$PWD/gpurun_out/r5s
mkdir -p $O
PCONV_HIP_LIB=$PWD/tools/_build/libpconv_hip_ee4stamp.so timeout -k 10 200 python tools/gpu_probe_ee4_stamps.py 2>&1 | tee $O/stamps.txt
