#!/bin/bash
# round 4, call H: F(4x2,3x3) as the default for the 64-multiple-cout layers -- parity of the whole codec, bench
set -o pipefail
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4h_smoke.log 2>&1 || { tail -30 gpurun_out/r4h_smoke.log; exit 1; }
tail -1 gpurun_out/r4h_smoke.log
timeout -k 10 900 python -m pytest tests/test_gpu_wino42.py tests/test_gpu_wino.py tests/test_gpu_codec_vs_oracle.py tests/test_reference_graph.py -m gpu -x -q --durations=5 > gpurun_out/r4h_tests.log 2>&1 || { tail -50 gpurun_out/r4h_tests.log; exit 1; }
tail -9 gpurun_out/r4h_tests.log
PCONV_BENCH_TABLE=1 timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r4h_bench.json 2> gpurun_out/r4h_bench.err || { tail -20 gpurun_out/r4h_bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4h_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'])
PY
