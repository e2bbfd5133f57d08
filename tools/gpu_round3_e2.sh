#!/bin/bash
# round 3: bulk entropy (encoder) kernel with the bias / slope table in LDS: parity tests + encode timings
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_codec_vs_oracle.py -m gpu -x -q -k "not metric_size" > gpurun_out/r3e2_pytest.log 2>&1 || { tail -40 gpurun_out/r3e2_pytest.log; exit 1; }
tail -2 gpurun_out/r3e2_pytest.log
timeout -k 10 300 python tools/gpu_probe_engine.py --batch8 2>gpurun_out/r3e2_err.log | grep "2048x4096" > gpurun_out/r3e2_engine.txt
cat gpurun_out/r3e2_engine.txt
