#!/bin/bash
# rocprofv3 PMC passes (separate runs, counters only) over the dense tile convolution probe
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_conv_$i
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_conv_$i -- python3 $R/tools/gpu_probe_conv.py 192 192 3 1 64 2048 3 > $R/gpurun_out/pmc_conv_$i.log 2>&1 || tail -3 $R/gpurun_out/pmc_conv_$i.log
done
echo done
