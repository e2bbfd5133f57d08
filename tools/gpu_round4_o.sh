#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_codec_vs_oracle.py -m gpu -x -q > gpurun_out/r4o_tests.log 2>&1 || { tail -40 gpurun_out/r4o_tests.log; exit 1; }
tail -3 gpurun_out/r4o_tests.log
for i in 1 2; do PCONV_BENCH_TABLE=1 timeout -k 10 400 python bench.py --no-cpu-baseline --steps 3 2>/dev/null | cut -c1-160; done
