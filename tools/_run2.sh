set -e
R=$GRAFT_REPO_ROOT
python tools/gpu_probe_conv.py
python tools/gpu_probe_conv.py 192 192 3 1 32 1024 10
python tools/gpu_probe_conv.py 192 768 3 1 32 1024 5
python tools/gpu_probe_conv.py 96 96 3 1 32 1024 10
python tools/gpu_probe_conv.py 192 96 1 1 32 1024 10
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_conv1 -- python3 $R/tools/gpu_probe_conv.py > $R/gpurun_out/pmc_conv1.log 2>&1 || tail -5 $R/gpurun_out/pmc_conv1.log
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc_conv2 -- python3 $R/tools/gpu_probe_conv.py > $R/gpurun_out/pmc_conv2.log 2>&1 || tail -5 $R/gpurun_out/pmc_conv2.log
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmc_conv3 -- python3 $R/tools/gpu_probe_conv.py > $R/gpurun_out/pmc_conv3.log 2>&1 || tail -5 $R/gpurun_out/pmc_conv3.log
echo done
