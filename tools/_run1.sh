set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/pytest_gpu.log
python bench.py > gpurun_out/bench_r1c.json 2> gpurun_out/bench_r1c.err || { tail -30 gpurun_out/bench_r1c.err; exit 1; }
cat gpurun_out/bench_r1c.json
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r1c
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1c -- python3 $R/bench.py --steps 1 --warmup 0 --prime 1 --no-cpu-baseline > $R/gpurun_out/prof_r1c.log 2>&1 || tail -5 $R/gpurun_out/prof_r1c.log
echo done
