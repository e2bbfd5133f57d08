#!/bin/bash
# round 3, session E: Winograd tile convolution -- parity tests, then timings against the direct kernel
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_wino.py -m gpu -x -q > gpurun_out/r3e_pytest.log 2>&1 || { tail -60 gpurun_out/r3e_pytest.log; exit 1; }
tail -3 gpurun_out/r3e_pytest.log
timeout -k 10 300 python tools/gpu_probe_wino.py > gpurun_out/r3e_wino.txt 2>gpurun_out/r3e_wino.err || { tail -20 gpurun_out/r3e_wino.err; exit 1; }
cat gpurun_out/r3e_wino.txt
