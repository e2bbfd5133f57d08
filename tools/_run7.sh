set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_f $R/gpurun_out/pmc_w
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_f -- python3 $R/tools/gpu_probe_conv.py 192 192 3 1 64 2048 3 > $R/gpurun_out/pmc_f.log 2>&1 || tail -5 $R/gpurun_out/pmc_f.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_w -- python3 $R/tools/gpu_probe_conv.py 192 192 3 1 64 2048 3 > $R/gpurun_out/pmc_w.log 2>&1 || tail -5 $R/gpurun_out/pmc_w.log
echo done
