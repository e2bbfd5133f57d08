#!/bin/bash
# Round 5, L: the input layer (14 channels) on the matrix cores too; the constrained host plan as a GPU test.
O=$PWD/gpurun_out/r5l
mkdir -p $O
R=$PWD
timeout -k 10 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_entropy_mfma.py tests/test_gpu_codec_vs_oracle.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
for b0 in valu mfma; do
  echo "== PCONV_EE_BULK0=$b0"
  PCONV_EE_BULK0=$b0 python tools/gpu_probe_entropy_mfma.py 8 3 16 512 2>&1 | grep -v amdgpu.ids
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_t && PCONV_EE_BULK0=$b0 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -- python3 $R/tools/gpu_probe_entropy_mfma.py 2 1 16 512 > $O/trace_$b0.log 2>&1
    python3 - <<PY
import csv, glob
f = glob.glob('/tmp/prof_t/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'bulk' in r['Kernel_Name'] and ('<14' in r['Kernel_Name'])]
d = {}
for r in rows:
    k = ('mfma14' if 'mfma' in r['Kernel_Name'] else 'valu14', r['Grid_Size_X'])
    d.setdefault(k, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items()):
    print(k, 'launches', len(v), 'avg us %.1f' % (sum(v) / len(v)), 'min %.1f max %.1f' % (min(v), max(v)))
PY
  )
done 2>&1 | tee $O/layer0.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'MPix/s', d['value'], 'ms/step', d['ms_per_step'])"; }
for rep in 1 2; do for b0 in valu mfma; do PCONV_EE_BULK0=$b0 timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check 2>/dev/null | line "bench [layer 0 $b0] rep $rep:"; done; done | tee $O/bench.txt
