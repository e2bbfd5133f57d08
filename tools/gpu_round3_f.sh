#!/bin/bash
# round 3, session F: whole GPU suite with the Winograd default, then the bench line
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r3f_pytest.log 2>&1 || { tail -60 gpurun_out/r3f_pytest.log; exit 1; }
tail -14 gpurun_out/r3f_pytest.log
PCONV_BENCH_TABLE=1 timeout -k 10 400 python bench.py > gpurun_out/r3f_bench.json 2> gpurun_out/r3f_bench.err || { tail -20 gpurun_out/r3f_bench.err; exit 1; }
cut -c1-600 gpurun_out/r3f_bench.json
