#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_codec_vs_oracle.py -x -q -m gpu > $O/r4aa_tests.txt 2>&1; rc=$?
tail -4 $O/r4aa_tests.txt
[ $rc = 0 ] || exit 1
PCONV_BENCH_TABLE=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/r4aa_bench.json 2> $O/r4aa_bench.err || { tail -5 $O/r4aa_bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4aa_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["tile_conv_s_per_step"])
tot={}
for r in d["roofline_table"]:
    k=r["kernel"].split("<")[0]+("" if "conv_mfma" not in r["kernel"] else " "+r["class"].split(" ")[0]+r["class"].split(" ")[1])
    tot[k]=tot.get(k,0)+r["launches"]*r["avg_launch_ms"]/d["steps"]
for k,v in sorted(tot.items(), key=lambda kv:-kv[1]): print("%-40s %7.1f ms/step" % (k,v))
for r in d["roofline_table"]:
    if "s2" in r["class"] or "->12" in r["class"]:
        print("%-26s %-40s n=%3d %8.3f ms  mfma %.3f  hbm %s" % (r["class"], r["kernel"][:40], r["launches"], r["avg_launch_ms"], r["frac"], r.get("hbm_frac")))
PY
