#!/bin/bash
# session E: 3-stage DMA pipeline of the tiled 1x1 kernel -- parity, then per-scale table
set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -m gpu -x -q > gpurun_out/r2e_pytest.log 2>&1 || { tail -40 gpurun_out/r2e_pytest.log; exit 1; }
grep -q "Memory access fault" gpurun_out/r2e_pytest.log && exit 1
tail -3 gpurun_out/r2e_pytest.log
for mode in tiled resident; do
PCONV_CONV1X1=$mode python bench.py --mode analysis --height 2048 --width 4096 --steps 3 --warmup 1 > gpurun_out/r2e_analysis_$mode.json 2>/dev/null
python - <<PY
import json
d=json.load(open('gpurun_out/r2e_analysis_$mode.json'))
print('$mode', d['value'], d['ms_per_step'])
for r in d['roofline_table']:
    print('  ', r['class'], r['launches'], r['avg_launch_ms'], r['achieved'], r['frac'])
PY
done
echo done
