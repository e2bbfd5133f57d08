#!/bin/bash
# session G: register-blocked resident 1x1 kernel (parity, A/B table); single-poller queued chain timings
set -e
mkdir -p gpurun_out
timeout -k 10 300 python tools/gpu_diag_1x1.py 16 192 34 1026 96 > gpurun_out/r2g_diag.log 2>&1 || { tail -5 gpurun_out/r2g_diag.log; exit 1; }
grep -q "Memory access fault" gpurun_out/r2g_diag.log && exit 1
tail -2 gpurun_out/r2g_diag.log
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q > gpurun_out/r2g_pytest.log 2>&1 || { tail -40 gpurun_out/r2g_pytest.log; exit 1; }
grep -q "Memory access fault" gpurun_out/r2g_pytest.log && exit 1
tail -3 gpurun_out/r2g_pytest.log
for mode in auto tiled; do
PCONV_CONV1X1=$mode python bench.py --mode analysis --height 2048 --width 4096 --steps 3 --warmup 1 > gpurun_out/r2g_analysis_$mode.json 2>/dev/null
python - <<PY
import json
d=json.load(open('gpurun_out/r2g_analysis_$mode.json'))
print('$mode', d['value'], d['ms_per_step'])
for r in d['roofline_table']:
    if r['class'].startswith('1x1') or r['class'].startswith('GDN'): print('  ', r['class'], r['kernel'][:24], r['launches'], r['avg_launch_ms'], r['achieved'], r['frac'])
PY
done
PCONV_ENGINE_TIMING=1 timeout -k 10 300 python tools/gpu_probe_engine.py --batch --batch8 > gpurun_out/r2g_probe_engine.log 2>&1 || { tail -5 gpurun_out/r2g_probe_engine.log; exit 1; }
grep "rep1\|decode" gpurun_out/r2g_probe_engine.log
echo done
