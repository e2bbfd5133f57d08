set -e
python tools/gpu_probe_engine.py --batch 2>&1 | grep -v rep0
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_eng
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_eng -- python3 $R/tools/gpu_probe_engine.py --batch > $R/gpurun_out/prof_eng.log 2>&1 || tail -5 $R/gpurun_out/prof_eng.log
