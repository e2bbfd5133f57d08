#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_codec_vs_oracle.py -x -q -m gpu > $O/r4y_tests.txt 2>&1; rc=$?
tail -4 $O/r4y_tests.txt
[ $rc = 0 ] || exit 1
PCONV_BENCH_TABLE=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/r4y_bench.json 2> $O/r4y_bench.err || { tail -5 $O/r4y_bench.err; exit 1; }
PCONV_CONV1X1_WAYOUT=pipe python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/r4y_bench_pipe.json 2> $O/r4y_bench_pipe.err || { tail -5 $O/r4y_bench_pipe.err; exit 1; }
python - <<'PY'
import json
for f in ("r4y_bench.json", "r4y_bench_pipe.json"):
    d=json.loads(open("gpurun_out/"+f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["config"]["tile_conv_s_per_step"])
d=json.loads(open("gpurun_out/r4y_bench.json").read().strip().splitlines()[-1])
for r in d["roofline_table"]:
    if "s2" in r["class"]:
        print("%-26s n=%3d %8.3f ms  mfma %.3f  hbm %s" % (r["class"], r["launches"], r["avg_launch_ms"], r["frac"], r.get("hbm_frac")))
PY
