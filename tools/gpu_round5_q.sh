#!/bin/bash
# Round 5, Q: matrix-pipe busy share of the four-block encoder kernel and of its timing ablations (PMC pass, no trace)
O=$PWD/gpurun_out/r5q
mkdir -p $O
export TMPDIR=/tmp
for v in ${VARIANTS:-base ee4abl14}; do
  lib=$GRAFT_REPO_ROOT/tools/_build/libpconv_hip_$v.so
  [ $v = base ] && lib=$GRAFT_REPO_ROOT/pseudocylindrical_convolution_amd/libpconv_hip.so
  ( cd /tmp && PCONV_HIP_LIB=$lib PCONV_ENGINE_ENCODE_RANGES=1 timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_$v -o p -- python3 $GRAFT_REPO_ROOT/tools/gpu_probe_entropy_mfma.py 1 2 > $O/pmc_$v.log 2>&1 )
  f=$(ls $O/pmc_$v/*/p_counter_collection.csv $O/pmc_$v/p_counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 - $f $v <<'PY' | tee -a $O/busy.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if 'ee_conv_bulk_mfma' not in r['Kernel_Name'] or int(r['Grid_Size']) < 1000000: continue
    k = r['Kernel_Name'].split('(')[0][-40:]
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k, c in acc.items():
    per = {x: c[x] / n[(k, x)] for x in c}
    print(sys.argv[2], k, {x: round(v) for x, v in per.items()}, 'mfma_busy %.3f' % (per['SQ_VALU_MFMA_BUSY_CYCLES'] / (per['SQ_BUSY_CYCLES'] / 32 * 1024)))
PY
done
exit 0
