set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/pytest_gpu.log
python __graft_entry__.py --smoke 2>&1 | tail -1
python bench.py > gpurun_out/bench_r1_final.json 2> gpurun_out/bench_r1_final.err || { tail -30 gpurun_out/bench_r1_final.err; exit 1; }
cat gpurun_out/bench_r1_final.json
python bench.py --frames-per-gpu 1 --no-cpu-baseline > gpurun_out/bench_r1_f1.json 2>/dev/null; cat gpurun_out/bench_r1_f1.json | cut -c1-200
python bench.py --frames-per-gpu 4 --no-cpu-baseline > gpurun_out/bench_r1_f4.json 2>/dev/null; cat gpurun_out/bench_r1_f4.json | cut -c1-200
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_final
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -- python3 $R/bench.py --steps 1 --warmup 0 --prime 1 --no-cpu-baseline > $R/gpurun_out/prof_final.json 2> $R/gpurun_out/prof_final.err || tail -5 $R/gpurun_out/prof_final.err
cat $R/gpurun_out/prof_final.json | cut -c1-200
echo done
