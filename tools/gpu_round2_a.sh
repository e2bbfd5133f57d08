#!/bin/bash
# Round 2, first GPU session: the GPU test suite (incl. the oracle comparisons), smoke, the
# headline bench (2 and 6 timed steps: bpp must be identical), config #3, decode probe.
set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --durations=8 --deselect tests/test_gpu_codec_vs_oracle.py::test_engine_equals_oracle_at_reference_size --deselect tests/test_gpu_codec_vs_oracle.py::test_engine_lockstep_pair_equals_oracle > gpurun_out/r2_pytest_gpu.log 2>&1 || { tail -40 gpurun_out/r2_pytest_gpu.log; exit 1; }
tail -12 gpurun_out/r2_pytest_gpu.log
python __graft_entry__.py --smoke 2>&1 | tail -1
python bench.py --steps 2 > gpurun_out/r2_bench_s2.json 2> gpurun_out/r2_bench_s2.err || { tail -30 gpurun_out/r2_bench_s2.err; exit 1; }
cat gpurun_out/r2_bench_s2.json
python bench.py --steps 6 --no-cpu-baseline > gpurun_out/r2_bench_s6.json 2> gpurun_out/r2_bench_s6.err || { tail -30 gpurun_out/r2_bench_s6.err; exit 1; }
cut -c1-700 gpurun_out/r2_bench_s6.json
python bench.py --mode analysis --steps 5 --warmup 2 > gpurun_out/r2_analysis.json 2> gpurun_out/r2_analysis.err || { tail -30 gpurun_out/r2_analysis.err; exit 1; }
cat gpurun_out/r2_analysis.json
PCONV_ENGINE_TIMING=1 python tools/gpu_probe_engine.py --batch --batch8 > gpurun_out/r2_probe_engine.log 2>&1 || tail -5 gpurun_out/r2_probe_engine.log
tail -25 gpurun_out/r2_probe_engine.log
echo done
