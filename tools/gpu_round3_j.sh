#!/bin/bash
set -e
mkdir -p gpurun_out
PCONV_BENCH_TABLE=1 timeout -k 10 400 python bench.py --no-cpu-baseline --steps 3 > gpurun_out/r3j_bench.json 2> gpurun_out/r3j_bench.err || { tail -20 gpurun_out/r3j_bench.err; exit 1; }
cut -c1-300 gpurun_out/r3j_bench.json
