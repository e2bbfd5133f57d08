#!/bin/bash
# Round 2, session B: XCD-order A/B on the 3x3 probe, PMC evidence for the conv classes
# (analysis transform at 4096x2048) and for the decoder's step kernel.
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
echo "== conv probe, plain order" ; PCONV_CONV_XCD=0 python3 $R/tools/gpu_probe_conv.py 192 192 3 1 64 2048 5 2>&1 | tail -2
echo "== conv probe, XCD order"   ; PCONV_CONV_XCD=1 python3 $R/tools/gpu_probe_conv.py 192 192 3 1 64 2048 5 2>&1 | tail -2
BENCH="python3 $R/bench.py --mode analysis --height 2048 --width 4096 --steps 1 --warmup 0 --prime 1"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pmc_an_$i
  rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_an_$i -- $BENCH > $R/gpurun_out/r2_pmc_an_$i.log 2>&1 || { tail -5 $R/gpurun_out/r2_pmc_an_$i.log; }
done
python3 $R/tools/summarise_pmc.py $R/gpurun_out/r2_pmc_analysis.json /tmp/pmc_an_1 /tmp/pmc_an_2 /tmp/pmc_an_3 /tmp/pmc_an_4
# XCD order off, FETCH only: how much of the re-read goes away
rm -rf /tmp/pmc_an_x; PCONV_CONV_XCD=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_an_x -- $BENCH > $R/gpurun_out/r2_pmc_an_x.log 2>&1 || true
python3 $R/tools/summarise_pmc.py $R/gpurun_out/r2_pmc_analysis_plain_order.json /tmp/pmc_an_x
# decoder step kernel: one group of 4 frames, and one frame
export PCONV_ENGINE_GROUPS=1
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1)); rm -rf /tmp/pmc_ee_$i
  timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_ee_$i -- python3 $R/tools/gpu_probe_entropy.py --n=4 --once > $R/gpurun_out/r2_pmc_ee_$i.log 2>&1 || { tail -5 $R/gpurun_out/r2_pmc_ee_$i.log; }
  tail -1 $R/gpurun_out/r2_pmc_ee_$i.log
done
python3 $R/tools/summarise_pmc.py $R/gpurun_out/r2_pmc_entropy_n4.json /tmp/pmc_ee_1 /tmp/pmc_ee_2 /tmp/pmc_ee_3
echo done
