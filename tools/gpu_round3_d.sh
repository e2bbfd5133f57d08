#!/bin/bash
# round 3, session D: reference-graph fixture on the HIP path; deeper transform splits; default bench line
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_reference_graph.py -m gpu -q > gpurun_out/r3d_pytest.log 2>&1 || { tail -60 gpurun_out/r3d_pytest.log; }
tail -5 gpurun_out/r3d_pytest.log
OUT=gpurun_out/r3d_split.txt
: > $OUT
for cfg in "3 8" "2 9" "1 10" "3 8" "2 9"; do
  set -- $cfg
  echo "== analysis_split $1 synthesis_split $2" >> $OUT
  PCONV_ANALYSIS_SPLIT=$1 PCONV_SYNTHESIS_SPLIT=$2 timeout -k 10 300 python bench.py --steps 3 --no-cpu-baseline 2>>gpurun_out/r3d_err.log \
    | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], 'MPix/s', d['ms_per_step'], 'ms/step', 'conv_s', d['config']['tile_conv_s_per_step'])" >> $OUT
  tail -1 $OUT
done
PCONV_BENCH_TABLE=1 timeout -k 10 400 python bench.py > gpurun_out/r3d_bench.json 2> gpurun_out/r3d_bench.err || { tail -20 gpurun_out/r3d_bench.err; exit 1; }
cut -c1-400 gpurun_out/r3d_bench.json
echo done
