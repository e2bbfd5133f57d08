#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_codec_vs_oracle.py -m gpu -x -q -k "not metric_size" > gpurun_out/r3o_pytest.log 2>&1 || { tail -40 gpurun_out/r3o_pytest.log; exit 1; }
tail -3 gpurun_out/r3o_pytest.log
timeout -k 10 300 python tools/gpu_probe_engine.py --batch --batch8 2>/dev/null | grep rep1 | cut -c1-110
timeout -k 10 300 python bench.py --steps 3 --no-cpu-baseline 2>/dev/null | cut -c1-160
