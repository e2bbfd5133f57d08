#!/bin/bash
# round 4, call B: the band kernels (causal-compact order) -- parity first, then timing
set -o pipefail
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4b_smoke.log 2>&1 || { tail -30 gpurun_out/r4b_smoke.log; exit 1; }
tail -1 gpurun_out/r4b_smoke.log
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py tests/test_gpu_wino.py -m gpu -x -q --durations=8 > gpurun_out/r4b_tests1.log 2>&1 || { tail -60 gpurun_out/r4b_tests1.log; exit 1; }
tail -14 gpurun_out/r4b_tests1.log
timeout -k 10 900 python -m pytest tests/test_gpu_codec_vs_oracle.py -m gpu -x -q --durations=8 > gpurun_out/r4b_tests2.log 2>&1 || { tail -60 gpurun_out/r4b_tests2.log; exit 1; }
tail -14 gpurun_out/r4b_tests2.log
PCONV_ENGINE_TIMING=1 timeout -k 10 300 python tools/gpu_probe_engine.py --batch --batch8 > gpurun_out/r4b_engine.txt 2>&1 || { tail -20 gpurun_out/r4b_engine.txt; exit 1; }
cat gpurun_out/r4b_engine.txt | tail -30
PCONV_BENCH_TABLE=1 timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r4b_bench.json 2> gpurun_out/r4b_bench.err || { tail -20 gpurun_out/r4b_bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4b_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'])
PY
