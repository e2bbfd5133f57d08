#!/bin/bash
# round 3, session N: final validation -- whole GPU suite, smoke, default bench line (now with the PMC fields)
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=6 > gpurun_out/r3n_pytest.log 2>&1 || { tail -60 gpurun_out/r3n_pytest.log; exit 1; }
tail -12 gpurun_out/r3n_pytest.log
python __graft_entry__.py --smoke 2>&1 | tail -1
python bench.py > gpurun_out/r3n_bench.json 2> gpurun_out/r3n_bench.err || { tail -20 gpurun_out/r3n_bench.err; exit 1; }
cat gpurun_out/r3n_bench.json | cut -c1-2500
