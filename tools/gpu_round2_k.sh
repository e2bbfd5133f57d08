#!/bin/bash
# session K: pipelined encode + faster coder -- engine parity, bench
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_codec_vs_oracle.py tests/test_coder.py -m "gpu or not gpu" -x -q > gpurun_out/r2k_pytest.log 2>&1 || { tail -40 gpurun_out/r2k_pytest.log; exit 1; }
grep -q "Memory access fault" gpurun_out/r2k_pytest.log && exit 1
tail -3 gpurun_out/r2k_pytest.log
python __graft_entry__.py --smoke 2>&1 | tail -1
PCONV_ENGINE_TIMING=1 python bench.py --steps 3 --no-cpu-baseline > gpurun_out/r2k_bench.json 2> gpurun_out/r2k_bench.err || { tail -30 gpurun_out/r2k_bench.err; exit 1; }
cut -c1-600 gpurun_out/r2k_bench.json
grep "encode\|decode" gpurun_out/r2k_bench.err | tail -12
for f in 1 2 4; do python bench.py --steps 3 --no-cpu-baseline --frames-per-gpu $f 2>/dev/null | cut -c1-140; done
echo done
