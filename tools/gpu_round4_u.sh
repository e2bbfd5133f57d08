#!/bin/bash
PCONV_HIP_LIB=$PWD/tools/_build/libpconv_hip_sstamp.so PCONV_CONV1X1=stream timeout -k 10 200 python tools/gpu_probe_stream_stamps.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r4u_stream_stamps.txt
