#!/bin/bash
PCONV_HIP_LIB=$PWD/tools/_build/libpconv_hip_w42stamp.so timeout -k 10 120 python tools/gpu_probe_wino42_stamps.py 2>&1 | tee gpurun_out/r4j_w42_stamps.txt
