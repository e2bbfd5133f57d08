#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_wino.py tests/test_gpu_engine.py -m gpu -x -q > gpurun_out/r3l_pytest.log 2>&1 || { tail -40 gpurun_out/r3l_pytest.log; exit 1; }
tail -3 gpurun_out/r3l_pytest.log
timeout -k 10 600 python -m pytest tests/test_gpu_codec_vs_oracle.py -m gpu -x -q -k "reference_size or lockstep or config2 or config3" > gpurun_out/r3l_pytest2.log 2>&1 || { tail -40 gpurun_out/r3l_pytest2.log; exit 1; }
tail -3 gpurun_out/r3l_pytest2.log
PCONV_BENCH_TABLE=1 timeout -k 10 400 python bench.py --no-cpu-baseline --steps 3 > gpurun_out/r3l_bench.json 2> gpurun_out/r3l_bench.err || { tail -20 gpurun_out/r3l_bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3l_bench.json').readline())
print(d['value'], d['ms_per_step'], d['config']['tile_conv_s_per_step'], d['roofline'])
for r in d['hbm']: print(r['kernel'], r['launches'], r['avg_launch_us'], r['achieved'])
PY
