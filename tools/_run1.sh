set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r1d
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1d -- python3 $R/bench.py --steps 1 --warmup 0 --prime 1 --no-cpu-baseline --frames-per-gpu 4 > $R/gpurun_out/prof_r1d.log 2>&1 || tail -5 $R/gpurun_out/prof_r1d.log
echo done
