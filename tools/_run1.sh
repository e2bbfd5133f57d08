set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/pytest_gpu.log
python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1 || { tail -30 gpurun_out/smoke.log; exit 1; }
tail -2 gpurun_out/smoke.log
python bench.py > gpurun_out/bench_r1b.json 2> gpurun_out/bench_r1b.err || { tail -30 gpurun_out/bench_r1b.err; exit 1; }
cat gpurun_out/bench_r1b.json
PCONV_SKIP_DEAD=0 python bench.py --no-cpu-baseline > gpurun_out/bench_r1b_noskip.json 2> gpurun_out/bench_r1b_noskip.err
cat gpurun_out/bench_r1b_noskip.json
python tools/gpu_probe_conv.py > gpurun_out/conv_probe.log 2>&1
cat gpurun_out/conv_probe.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_conv1 -- python3 $R/tools/gpu_probe_conv.py > $R/gpurun_out/pmc_conv1.log 2>&1 || tail -5 $R/gpurun_out/pmc_conv1.log
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc_conv2 -- python3 $R/tools/gpu_probe_conv.py > $R/gpurun_out/pmc_conv2.log 2>&1 || tail -5 $R/gpurun_out/pmc_conv2.log
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmc_conv3 -- python3 $R/tools/gpu_probe_conv.py > $R/gpurun_out/pmc_conv3.log 2>&1 || tail -5 $R/gpurun_out/pmc_conv3.log
echo done
