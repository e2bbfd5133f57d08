#!/bin/bash
# round 3: step kernel with the gather offsets of masked entries folded onto the window origin: parity + decode timings
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_codec_vs_oracle.py -m gpu -x -q -k "not metric_size" > gpurun_out/r3e3_pytest.log 2>&1 || { tail -40 gpurun_out/r3e3_pytest.log; exit 1; }
tail -2 gpurun_out/r3e3_pytest.log
timeout -k 10 300 python tools/gpu_probe_engine.py --batch --batch8 2>gpurun_out/r3e3_err.log | grep "2048x4096" | grep rep1 > gpurun_out/r3e3_engine.txt
cat gpurun_out/r3e3_engine.txt | cut -c1-150
PCONV_ENGINE_TRACE=1 timeout -k 10 300 python bench.py --steps 3 --no-cpu-baseline 2>gpurun_out/r3e3_bench.err | cut -c1-200
grep "decode 8 frame" gpurun_out/r3e3_bench.err | tail -2
