#!/bin/bash
# Round 5, D: where the matrix-core encoder kernel's time goes: per-dispatch trace (full / ranged launches), PMC passes;
# the 16x16x1_4b probe; the part-owner flag probe; the repaired clip test.
O=$PWD/gpurun_out/r5d
mkdir -p $O
R=$PWD
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "leaky" 2>&1 | tail -3
./tools/_build/mfma16x1_4b_probe 2>&1 | tee $O/mfma16x1_4b_probe.txt
timeout -k 10 600 ./tools/_build/flag_chain_probe 780 1 2>&1 | tee $O/flag_chain_owner.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_t && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -- python3 $R/tools/gpu_probe_entropy_mfma.py 2 1 16 512 > $O/trace.log 2>&1
python3 - <<PY > $O/mfma_dispatches.txt
import csv, glob
f = glob.glob('/tmp/prof_t/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'bulk' in r['Kernel_Name']]
print(list(rows[0].keys()))
for r in rows:
    print(r['Kernel_Name'][27:60], int(r['End_Timestamp']) - int(r['Start_Timestamp']), 'ns grid', r.get('Grid_Size_X'), r.get('Grid_Size_Y'), 'lds', r.get('LDS_Block_Size'), 'vgpr', r.get('VGPR_Count'), 'accum', r.get('Accum_VGPR_Count'), 'scratch', r.get('Scratch_Size'))
PY
head -40 $O/mfma_dispatches.txt | cut -c1-200
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1)); rm -rf /tmp/pmc_d_$i
  PCONV_ENGINE_ENCODE_RANGES=1 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_d_$i -- python3 $R/tools/gpu_probe_entropy_mfma.py 1 1 16 512 > $O/pmc_$i.log 2>&1
  python3 - <<PY >> $O/mfma_pmc.txt
import csv, glob, collections
f = glob.glob('/tmp/pmc_d_$i/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen=set()
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name']
    if 'bulk' not in k: continue
    name = 'mfma' if 'mfma' in k else ('valu42' if '<42' in k else 'valu14')
    acc[name][r['Counter_Name']] += float(r['Counter_Value'])
    if (r['Dispatch_Id']) not in seen: seen.add(r['Dispatch_Id']); n[name]+=1
for name in acc:
    print(name, 'dispatches', n[name], {c: round(v / max(n[name],1)) for c, v in acc[name].items()})
PY
done
cat $O/mfma_pmc.txt
