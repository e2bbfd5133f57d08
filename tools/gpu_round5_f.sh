#!/bin/bash
# Round 5, F: pipelined operand reads + ranges only on the last chunk: variants, whole GPU suite, bench A/B, split-bf16 probe.
O=$PWD/gpurun_out/r5f
mkdir -p $O
R=$PWD
./tools/_build/bf16x3_probe 2>&1 | tee $O/bf16x3_probe.txt
for cfg in "1 4" "2 4"; do
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
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee $O/gpu_tests.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$1', 'MPix/s', d['value'], 'ms/step', d['ms_per_step'], 'host_cores_busy', c['host_cores_busy'])"; }
for rep in 1 2; do
  for cfg in "valu" "mfma"; do
    PCONV_EE_BULK=$cfg PCONV_ENGINE_TIMING=1 timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check 2> $O/err.txt | line "bench [$cfg] rep $rep:"
  done
done 2>&1 | tee $O/bench.txt
for n in 1 2 4; do python bench.py --frames-per-gpu $n --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | cut -c1-140; done | tee $O/bench_frames_1_2_4.txt
