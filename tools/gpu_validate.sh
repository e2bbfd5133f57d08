#!/bin/bash
# On the GPU box (through gpurun): GPU test suite, smoke, the headline bench at 8 / 1 / 4 frames per
# GPU and a rocprofv3 kernel trace of one bench step; everything lands in gpurun_out/.
set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/pytest_gpu.log
python __graft_entry__.py --smoke 2>&1 | tail -1
python bench.py > gpurun_out/bench_r1_final.json 2> gpurun_out/bench_r1_final.err || { tail -30 gpurun_out/bench_r1_final.err; exit 1; }
cat gpurun_out/bench_r1_final.json
python bench.py --frames-per-gpu 1 --no-cpu-baseline > gpurun_out/bench_r1_f1.json 2>/dev/null; cut -c1-160 gpurun_out/bench_r1_f1.json
python bench.py --frames-per-gpu 4 --no-cpu-baseline > gpurun_out/bench_r1_f4.json 2>/dev/null; cut -c1-160 gpurun_out/bench_r1_f4.json
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_final
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -- python3 $R/bench.py --steps 1 --warmup 0 --prime 1 --no-cpu-baseline > $R/gpurun_out/prof_final.json 2> $R/gpurun_out/prof_final.err || tail -5 $R/gpurun_out/prof_final.err
cut -c1-160 $R/gpurun_out/prof_final.json
echo done
