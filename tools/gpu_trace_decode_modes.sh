#!/bin/bash
# kernel trace of an 8-frame entropy decode, queued vs host-driven chain: per-kernel durations and
# the busy time of the two group streams
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for m in queued host; do
  rm -rf /tmp/prof_$m
  PCONV_ENGINE_CHAIN=$m rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$m -- python3 $R/tools/gpu_probe_decode_modes.py 8 > $R/gpurun_out/trace_$m.log 2>&1 || tail -5 $R/gpurun_out/trace_$m.log
  python3 - $m <<'PY'
import csv, glob, sys, collections
m = sys.argv[1]
f = glob.glob('/tmp/prof_%s/**/*kernel_trace.csv' % m, recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if 'ee_' in r['Kernel_Name']]
# the last third of the trace = the last decode repetition
t0 = min(int(r['Start_Timestamp']) for r in rows); t1 = max(int(r['End_Timestamp']) for r in rows)
cut = t0 + (t1 - t0) * 2 // 3
rows = [r for r in rows if int(r['Start_Timestamp']) >= cut]
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    k = r['Kernel_Name'].split('(')[0][:60]
    agg[k][0] += 1; agg[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
span = (max(int(r['End_Timestamp']) for r in rows) - min(int(r['Start_Timestamp']) for r in rows)) / 1e6
print("== %s: %d kernels in %.1f ms" % (m, len(rows), span))
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:6]:
    print("  %-60s n=%6d avg %7.2f us total %7.1f ms" % (k, n, t / n / 1e3, t / 1e6))
# union of busy intervals
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows)
busy = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
print("  GPU busy (union of kernel intervals) %.1f ms, sum of durations %.1f ms" % (busy / 1e6, sum(e - s for s, e in iv) / 1e6))
PY
done
