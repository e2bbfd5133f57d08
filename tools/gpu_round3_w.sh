#!/bin/bash
# round 3, session W: direct kernel with the bias / slope table in LDS: ops tests, 1x1 / GDN probe old vs new, bench
set -e
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_wino.py -m gpu -x -q > gpurun_out/r3w_pytest.log 2>&1 || { tail -40 gpurun_out/r3w_pytest.log; exit 1; }
tail -2 gpurun_out/r3w_pytest.log
OUT=gpurun_out/r3w_1x1.txt
: > $OUT
for lib in v8 new v8 new; do
  echo "== $lib" >> $OUT
  if [ $lib = new ]; then unset PCONV_HIP_LIB; else export PCONV_HIP_LIB=$PWD/tools/_build/libpconv_hip_$lib.so; fi
  timeout -k 10 200 python tools/gpu_probe_1x1.py >> $OUT 2>gpurun_out/r3w_err.log || { tail -5 gpurun_out/r3w_err.log; exit 1; }
done
unset PCONV_HIP_LIB
cat $OUT
PCONV_BENCH_TABLE=1 timeout -k 10 400 python bench.py --no-cpu-baseline --steps 3 > gpurun_out/r3w_bench.json 2> gpurun_out/r3w_bench.err || { tail -20 gpurun_out/r3w_bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3w_bench.json').readline())
print(d['value'], d['ms_per_step'], d['config']['tile_conv_s_per_step'])
for r in d['roofline_table']: print(r['class'], r['launches'], r['avg_launch_ms'], r['achieved'])
PY
