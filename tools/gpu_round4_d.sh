#!/bin/bash
# round 4, call D: band kernels v2 -- parity (engine + per-op + oracle at two sizes), engine timings, kernel trace
set -o pipefail
mkdir -p gpurun_out
O=$PWD/gpurun_out
R=$PWD
timeout -k 10 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_ops.py -m gpu -x -q -k "engine or entropy" > $O/r4d_tests1.log 2>&1 || { tail -40 $O/r4d_tests1.log; exit 1; }
tail -3 $O/r4d_tests1.log
timeout -k 10 900 python -m pytest tests/test_gpu_codec_vs_oracle.py -m gpu -x -q -k "reference_size or lockstep" > $O/r4d_tests2.log 2>&1 || { tail -40 $O/r4d_tests2.log; exit 1; }
tail -3 $O/r4d_tests2.log
cd /tmp && export TMPDIR=/tmp
for n in 1 8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ee_$n -- python3 $R/tools/gpu_probe_entropy_only.py $n 3 > $O/r4d_ee_$n.txt 2> $O/r4d_ee_$n.err || { tail -5 $O/r4d_ee_$n.err; exit 1; }
  cat $O/r4d_ee_$n.txt
  f=$(find /tmp/prof_ee_$n -name "*kernel_stats.csv" | head -1)
  cp "$f" $O/r4d_ee_${n}_kernel_stats.csv
  head -7 "$f" | cut -c1-100,180-260
done
cd $R
for n in 1 2 4 8; do PCONV_ENGINE_TIMING=1 timeout -k 10 120 python tools/gpu_probe_entropy_only.py $n 2 2>&1 | grep -v "^\[pconv" | tail -1; done
