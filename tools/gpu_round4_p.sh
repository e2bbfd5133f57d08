#!/bin/bash
# kernel trace of one bench step -> busy / idle map
set -o pipefail
R=$PWD; O=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tr -- python3 $R/bench.py --steps 1 --warmup 0 --prime 2 --no-cpu-baseline --no-check > $O/r4p_bench.json 2> $O/r4p_bench.err || { tail -5 $O/r4p_bench.err; exit 1; }
f=$(find /tmp/prof_tr -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' > $O/r4p_last_step_trace.csv
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# keep the last step: find the last SphereSlice... simpler: the last 0.85 s of the trace
t1=max(int(r["End_Timestamp"]) for r in rows)
keep=[r for r in rows if int(r["Start_Timestamp"])>=t1-int(0.83e9)]
w=csv.DictWriter(sys.stdout, fieldnames=["Start_Timestamp","End_Timestamp","Kernel_Name"])
w.writeheader()
for r in keep: w.writerow({k:r[k] for k in ("Start_Timestamp","End_Timestamp","Kernel_Name")})
PY
cd $R
python tools/gpu_idle_map.py $O/r4p_last_step_trace.csv 12 20 | tee $O/r4p_idle_map.txt | cut -c1-150
