#!/bin/bash
# round 4, call E: PMC passes over the entropy engine's band kernels (counters only, separate passes)
set -o pipefail
R=$PWD
O=$PWD/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf $O/pmc_ee_$i
  rocprofv3 --pmc $set --output-format csv -d $O/pmc_ee_$i -- python3 $R/tools/gpu_probe_entropy_only.py 2 1 both > $O/pmc_ee_$i.log 2>&1 || { tail -3 $O/pmc_ee_$i.log; }
done
cd $R
python tools/summarise_pmc.py $O/r4e_pmc_ee.json $O/pmc_ee_1 $O/pmc_ee_2 $O/pmc_ee_3 > /dev/null 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4e_pmc_ee.json'))
for k,v in d['kernels'].items():
    if 'band' in k:
        print(k, {a: (round(b,1) if isinstance(b,float) else b) for a,b in v.items() if not isinstance(b,dict)})
        for a,b in v.items():
            if isinstance(b,dict): print('   ',a,{x:round(y,1) for x,y in b.items()})
PY
rm -rf $O/pmc_ee_1 $O/pmc_ee_2 $O/pmc_ee_3
