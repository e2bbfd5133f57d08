#!/bin/bash
# round 4, call A: the new tests (benchmarked workload, counted ties, relative Winograd error), the whole GPU
# suite, smoke, and the baseline bench line of the round
set -o pipefail
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4a_smoke.log 2>&1 || { tail -30 gpurun_out/r4a_smoke.log; exit 1; }
tail -1 gpurun_out/r4a_smoke.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r4a_tests.log 2>&1 || { tail -60 gpurun_out/r4a_tests.log; exit 1; }
tail -25 gpurun_out/r4a_tests.log
PCONV_BENCH_TABLE=1 timeout -k 10 400 python bench.py > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err || { tail -20 gpurun_out/r4a_bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4a_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'])
PY
