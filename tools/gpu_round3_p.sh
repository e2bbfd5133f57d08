#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_codec_vs_oracle.py -m gpu -x -q -k "not metric_size" > gpurun_out/r3p_pytest.log 2>&1 || { tail -40 gpurun_out/r3p_pytest.log; exit 1; }
tail -3 gpurun_out/r3p_pytest.log
OUT=gpurun_out/r3p_graph.txt
: > $OUT
for rep in 1 2; do
for cfg in "PCONV_ENGINE_GRAPH=0" "PCONV_ENGINE_GRAPH=1"; do
  echo "== rep $rep $cfg" >> $OUT
  env $cfg PCONV_ENGINE_TIMING=1 timeout -k 10 300 python bench.py --steps 3 --no-cpu-baseline 2>gpurun_out/r3p_err.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], 'MPix/s', d['ms_per_step'], 'ms/step')" >> $OUT
  grep "pconv engine\] decode" gpurun_out/r3p_err.log | tail -2 >> $OUT
done
done
for f in 4; do
for cfg in "PCONV_ENGINE_GRAPH=0" "PCONV_ENGINE_GRAPH=1"; do
  echo "== frames $f $cfg PCONV_ENGINE_CHAIN=host" >> $OUT
  env $cfg PCONV_ENGINE_CHAIN=host timeout -k 10 300 python bench.py --steps 3 --no-cpu-baseline --frames-per-gpu $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], 'MPix/s', d['ms_per_step'], 'ms/step')" >> $OUT
done
done
cat $OUT
