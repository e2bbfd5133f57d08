#!/bin/bash
# round 3, session I: PMC passes over the Winograd kernel (one shape)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/r3i_counters.txt 2>&1
grep -c . $R/gpurun_out/r3i_counters.txt
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_VMEM" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/r3i_pmc_$i
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/r3i_pmc_$i -- python3 $R/tools/gpu_probe_wino_one.py wino > $R/gpurun_out/r3i_pmc_$i.log 2>&1 || tail -2 $R/gpurun_out/r3i_pmc_$i.log
done
cd $R
python tools/summarise_pmc.py gpurun_out/r3i_pmc.json gpurun_out/r3i_pmc_* > /dev/null 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3i_pmc.json'))
for k,v in d.get('kernels',{}).items():
    if 'wino_conv' in k:
        print(k); print(json.dumps(v, indent=1)[:3000])
PY
