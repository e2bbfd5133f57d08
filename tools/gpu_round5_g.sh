#!/bin/bash
# Round 5, G: direct weight fetch (no LDS ring, no barrier in the class loop) against the ring form.
O=$PWD/gpurun_out/r5g
mkdir -p $O
R=$PWD
timeout -k 10 600 python -m pytest tests/test_gpu_entropy_mfma.py tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
for cfg in "direct 4" "ring 4" "direct 8"; do
  set -- $cfg
  export PCONV_EE_MFMA_WSRC=$1 PCONV_EE_MFMA_WAVES=$2
  echo "== weights $1, waves $2"
  python tools/gpu_probe_entropy_mfma.py 1 3 16 512 2>&1 | grep -v amdgpu.ids
  python tools/gpu_probe_entropy_mfma.py 8 3 16 512 2>&1 | grep -v amdgpu.ids
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_t && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -- python3 $R/tools/gpu_probe_entropy_mfma.py 2 1 16 512 > $O/trace_$1_$2.log 2>&1
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
  )
done 2>&1 | tee $O/variants.txt
unset PCONV_EE_MFMA_WSRC PCONV_EE_MFMA_WAVES
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$1', 'MPix/s', d['value'], 'ms/step', d['ms_per_step'], 'host_cores_busy', c['host_cores_busy'])"; }
for rep in 1 2; do
  for cfg in "ring" "direct"; do
    PCONV_EE_MFMA_WSRC=$cfg PCONV_ENGINE_TIMING=1 timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check 2> $O/err.txt | line "bench [$cfg] rep $rep:"
  done
done 2>&1 | tee $O/bench.txt
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1)); rm -rf /tmp/pmc_g_$i
  PCONV_ENGINE_ENCODE_RANGES=1 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_g_$i -- python3 $R/tools/gpu_probe_entropy_mfma.py 1 1 16 512 > $O/pmc_$i.log 2>&1
  python3 - <<PY >> $O/mfma_pmc.txt
import csv, glob, collections
f = glob.glob('/tmp/pmc_g_$i/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen=set()
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name']
    if 'bulk_mfma' not in k: continue
    name = 'mfma'
    acc[name][r['Counter_Name']] += float(r['Counter_Value'])
    if (r['Dispatch_Id']) not in seen: seen.add(r['Dispatch_Id']); n[name]+=1
for name in acc:
    print(name, 'dispatches', n[name], {c: round(v / max(n[name],1)) for c, v in acc[name].items()})
PY
done
cat $O/mfma_pmc.txt
