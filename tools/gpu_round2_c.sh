#!/bin/bash
# Round 2, session C: resident 1x1 kernel -- parity tests, then the analysis roofline table
set -e
mkdir -p gpurun_out
python tools/gpu_diag_1x1.py > gpurun_out/r2c_diag.log 2>&1 || { tail -5 gpurun_out/r2c_diag.log; exit 1; }
grep -q "Memory access fault" gpurun_out/r2c_diag.log && exit 1
tail -2 gpurun_out/r2c_diag.log
python -m pytest tests/test_gpu_ops.py -m gpu -x -q > gpurun_out/r2c_pytest.log 2>&1 || { tail -40 gpurun_out/r2c_pytest.log; exit 1; }
tail -3 gpurun_out/r2c_pytest.log
python bench.py --mode analysis --steps 5 --warmup 2 > gpurun_out/r2c_analysis.json 2> gpurun_out/r2c_analysis.err || { tail -30 gpurun_out/r2c_analysis.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2c_analysis.json'))
print(d['value'], d['ms_per_step'])
for r in d['roofline_table']: print(r['class'], r['kernel'][:40], r['launches'], r['avg_launch_ms'], r['achieved'], r['frac'])
PY
python bench.py --mode analysis --height 2048 --width 4096 --steps 3 --warmup 1 > gpurun_out/r2c_analysis_full.json 2> gpurun_out/r2c_analysis_full.err || { tail -30 gpurun_out/r2c_analysis_full.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2c_analysis_full.json'))
print(d['value'], d['ms_per_step'])
for r in d['roofline_table']: print(r['class'], r['kernel'][:40], r['launches'], r['avg_launch_ms'], r['achieved'], r['frac'])
PY
PCONV_CONV1X1=tiled python bench.py --mode analysis --height 2048 --width 4096 --steps 3 --warmup 1 > gpurun_out/r2c_analysis_full_tiled.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2c_analysis_full_tiled.json'))
print('tiled 1x1:', d['value'], d['ms_per_step'])
for r in d['roofline_table']: print(r['class'], r['kernel'][:40], r['launches'], r['avg_launch_ms'], r['achieved'], r['frac'])
PY
echo done
