#!/bin/bash
# kernel trace of the entropy probe (1 frame and batches): gaps between the dependent launches of a decode step
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_dec
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_dec -- python3 $R/tools/gpu_probe_engine.py > $R/gpurun_out/prof_dec.log 2>&1 || tail -5 $R/gpurun_out/prof_dec.log
tail -2 $R/gpurun_out/prof_dec.log
