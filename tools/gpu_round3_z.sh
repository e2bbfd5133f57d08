#!/bin/bash
# round 3, session Z: where the GPU idles inside a bench step (kernel trace of one step, union over streams)
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_idle
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_idle -- python3 $R/bench.py --steps 1 --warmup 0 --prime 1 --no-cpu-baseline > $R/gpurun_out/r3z_bench.json 2> $R/gpurun_out/r3z_bench.err || tail -5 $R/gpurun_out/r3z_bench.err
f=$(find /tmp/prof_idle -name "*kernel_trace.csv" | head -1)
python3 $R/tools/gpu_idle_map.py $f 10 20 > $R/gpurun_out/r3z_idle.txt; cp $f $R/gpurun_out/r3z_kernel_trace.csv
tail -5 $R/gpurun_out/r3z_idle.txt
cut -c1-200 $R/gpurun_out/r3z_bench.json
