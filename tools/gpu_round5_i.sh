#!/bin/bash
# Round 5, I: encoder kernel after the prologue / way-out trim (tests incl. the variants, timing); slice / uslice with 1 / 2 / 4 rows per workgroup.
O=$PWD/gpurun_out/r5i
mkdir -p $O
R=$PWD
timeout -k 10 600 python -m pytest tests/test_gpu_entropy_mfma.py tests/test_gpu_engine.py tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
python tools/gpu_probe_entropy_mfma.py 8 3 16 512 2>&1 | grep -v amdgpu.ids | tee $O/probe.txt
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_t && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -- python3 $R/tools/gpu_probe_entropy_mfma.py 2 1 16 512 > $O/trace.log 2>&1
    python3 - <<PY
import csv, glob
f = glob.glob('/tmp/prof_t/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'bulk_mfma' in r['Kernel_Name']]
d = {}
for r in rows:
    k = ('mfma', r['Grid_Size_X'])
    d.setdefault(k, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items()):
    print(k, 'launches', len(v), 'avg us %.1f' % (sum(v) / len(v)), 'min %.1f max %.1f' % (min(v), max(v)))
PY
) | tee -a $O/probe.txt
for rb in 4 2 1; do echo "== PCONV_RESAMPLE_ROWS=$rb"; PCONV_RESAMPLE_ROWS=$rb python tools/gpu_probe_hbm.py 2>&1 | grep -i "slice"; done | tee $O/resample_rows.txt
