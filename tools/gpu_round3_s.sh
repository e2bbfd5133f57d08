#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_engine.py -m gpu -x -q > gpurun_out/r3s_pytest.log 2>&1 || { tail -40 gpurun_out/r3s_pytest.log; exit 1; }
tail -3 gpurun_out/r3s_pytest.log
for f in 8 6 4; do PCONV_ENGINE_TIMING=1 timeout -k 10 300 python bench.py --steps 3 --no-cpu-baseline --frames-per-gpu $f 2>gpurun_out/r3s_err.log | cut -c1-150; grep "pconv engine\] decode" gpurun_out/r3s_err.log | tail -2 | head -1; done
