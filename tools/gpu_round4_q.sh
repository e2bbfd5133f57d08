#!/bin/bash
# coder fast loops on the box: engine timing lines + bench with the two-roof class table
set -o pipefail
R=$PWD; O=$PWD/gpurun_out
mkdir -p $O
python -m pytest tests/test_coder.py tests/test_gpu_engine.py -x -q -m "gpu or not gpu" > $O/r4q_tests.txt 2>&1 || { tail -20 $O/r4q_tests.txt; exit 1; }
tail -2 $O/r4q_tests.txt
PCONV_ENGINE_TIMING=1 PCONV_BENCH_TABLE=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/r4q_bench.json 2> $O/r4q_bench.err || { tail -5 $O/r4q_bench.err; exit 1; }
grep "pconv engine" $O/r4q_bench.err | tail -8
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4q_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
for r in d["roofline_table"]:
    print("%-26s %-52s n=%3d %8.3f ms  mfma %.3f  hbm %s" % (r["class"], r["kernel"][:52], r["launches"], r["avg_launch_ms"], r["frac"], r.get("hbm_frac")))
PY
