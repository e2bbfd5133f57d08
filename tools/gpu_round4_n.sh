#!/bin/bash
# round 4, call N: the bulk (encoder) kernel with slab DMA ring + LDS output tile: parity, timing
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_ops.py -m gpu -x -q -k "engine or entropy" > gpurun_out/r4n_tests1.log 2>&1 || { tail -40 gpurun_out/r4n_tests1.log; exit 1; }
tail -3 gpurun_out/r4n_tests1.log
timeout -k 10 900 python -m pytest tests/test_gpu_codec_vs_oracle.py -m gpu -x -q -k "reference_size or lockstep or reloaded" > gpurun_out/r4n_tests2.log 2>&1 || { tail -40 gpurun_out/r4n_tests2.log; exit 1; }
tail -3 gpurun_out/r4n_tests2.log
for n in 1 2 4 8; do PCONV_ENGINE_TIMING=1 timeout -k 10 120 python tools/gpu_probe_entropy_only.py $n 2 2>&1 | grep "rep1\|encode .* frame" | tail -2; done | tee gpurun_out/r4n_timing.txt
