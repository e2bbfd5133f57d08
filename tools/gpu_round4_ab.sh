#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_codec_vs_oracle.py -x -q -m gpu > $O/r4ab_tests.txt 2>&1; rc=$?
tail -4 $O/r4ab_tests.txt
[ $rc = 0 ] || exit 1
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/r4ab_bench.json 2> $O/r4ab_bench.err || { tail -5 $O/r4ab_bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4ab_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["tile_conv_s_per_step"])
for r in d["hbm"]:
    print("%-26s n=%4d %8.2f us  %7.1f GB/s  frac %.3f  | %s" % (r["kernel"], r["launches"], r["avg_launch_us"], r["achieved"], r["frac"], r["largest_class"]))
PY
