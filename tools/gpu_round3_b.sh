#!/bin/bash
# round 3, session B: whole GPU suite (new: metric-size engine-vs-oracle, reference-graph fixture, CLI end to end)
set -e
mkdir -p gpurun_out
nproc > gpurun_out/r3b_host.txt; free -g >> gpurun_out/r3b_host.txt
timeout -k 10 1050 python -m pytest tests -m gpu -x -q --durations=20 > gpurun_out/r3b_pytest.log 2>&1 || { tail -60 gpurun_out/r3b_pytest.log; exit 1; }
tail -30 gpurun_out/r3b_pytest.log
