#!/bin/bash
# Round 5, E: matrix-core encoder kernel after the address / range-list work: variants (rows per wave x waves per
# workgroup), per-dispatch durations of full and ranged launches, tests, bench against the vector kernel.
O=$PWD/gpurun_out/r5e
mkdir -p $O
R=$PWD
timeout -k 10 600 python -m pytest tests/test_gpu_entropy_mfma.py tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
for cfg in "2 4" "1 4" "2 8" "1 8"; do
  set -- $cfg
  export PCONV_EE_MFMA_NT=$1 PCONV_EE_MFMA_WAVES=$2
  echo "== rows per wave $1, waves $2"
  python tools/gpu_probe_entropy_mfma.py 1 3 16 512 2>&1 | grep -v amdgpu.ids
  python tools/gpu_probe_entropy_mfma.py 8 3 16 512 2>&1 | grep -v amdgpu.ids
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_t && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -- python3 $R/tools/gpu_probe_entropy_mfma.py 2 1 16 512 > $O/trace_$1_$2.log 2>&1
    python3 - <<PY
import csv, glob
f = glob.glob('/tmp/prof_t/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'bulk_mfma' in r['Kernel_Name'] or 'bulk_kernel<42' in r['Kernel_Name']]
d = {}
for r in rows:
    k = ('mfma' if 'mfma' in r['Kernel_Name'] else 'valu', r['Grid_Size_X'])
    d.setdefault(k, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items()):
    print(k, 'launches', len(v), 'avg us %.1f' % (sum(v) / len(v)), 'min %.1f max %.1f' % (min(v), max(v)))
PY
  )
done 2>&1 | tee $O/variants.txt
unset PCONV_EE_MFMA_NT PCONV_EE_MFMA_WAVES
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$1', 'MPix/s', d['value'], 'ms/step', d['ms_per_step'], 'host_cores_busy', c['host_cores_busy'])"; }
for rep in 1 2; do
  for cfg in "valu 2 4" "mfma 2 4" "mfma 1 4"; do
    set -- $cfg
    PCONV_EE_BULK=$1 PCONV_EE_MFMA_NT=$2 PCONV_EE_MFMA_WAVES=$3 PCONV_ENGINE_TIMING=1 timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check 2> $O/err.txt | line "bench [$cfg] rep $rep:"
    grep "encode 2" $O/err.txt | tail -1 | cut -c1-150
  done
done 2>&1 | tee $O/bench.txt
