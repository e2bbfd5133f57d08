#!/bin/bash
# Round 2, session D: step kernel v2 (records, DMA staging, pre-masked slabs) -- engine parity
# tests, decode timings; resident 1x1 without the serialised epilogue, per-scale table.
set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_engine.py tests/test_gpu_codec_vs_oracle.py -m gpu -x -q --deselect tests/test_gpu_codec_vs_oracle.py::test_engine_equals_oracle_at_reference_size > gpurun_out/r2d_pytest.log 2>&1 || { tail -40 gpurun_out/r2d_pytest.log; exit 1; }
grep -q "Memory access fault" gpurun_out/r2d_pytest.log && exit 1
tail -3 gpurun_out/r2d_pytest.log
python __graft_entry__.py --smoke 2>&1 | tail -1
PCONV_ENGINE_TIMING=1 python tools/gpu_probe_engine.py --batch --batch8 > gpurun_out/r2d_probe_engine.log 2>&1 || { tail -5 gpurun_out/r2d_probe_engine.log; exit 1; }
grep "rep1\|decode" gpurun_out/r2d_probe_engine.log
for mode in resident tiled; do
PCONV_CONV1X1=$mode python bench.py --mode analysis --height 2048 --width 4096 --steps 3 --warmup 1 > gpurun_out/r2d_analysis_$mode.json 2>/dev/null
python - <<PY
import json
d=json.load(open('gpurun_out/r2d_analysis_$mode.json'))
print('$mode', d['value'], d['ms_per_step'])
for r in d['roofline_table']:
    if r['class'].startswith('1x1') or r['class'].startswith('GDN'): print('  ', r['class'], r['launches'], r['avg_launch_ms'], r['achieved'], r['frac'])
PY
done
echo done
