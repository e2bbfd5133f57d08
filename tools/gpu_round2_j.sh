#!/bin/bash
# session J: backward kernels parity; XCD block order A/B on the transforms (same box, alternating)
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_backward.py -m gpu -x -q > gpurun_out/r2j_pytest.log 2>&1 || { tail -40 gpurun_out/r2j_pytest.log; exit 1; }
grep -q "Memory access fault" gpurun_out/r2j_pytest.log && exit 1
tail -3 gpurun_out/r2j_pytest.log
for rep in 1 2; do for x in 0 1; do
PCONV_CONV_XCD=$x python bench.py --mode analysis --height 2048 --width 4096 --steps 4 --warmup 1 > gpurun_out/r2j_an_x$x.json 2>/dev/null
python - <<PY
import json
d=json.load(open('gpurun_out/r2j_an_x$x.json'))
r=[t for t in d['roofline_table'] if t['class'].startswith('3x3 s1 192->192 w2048')][0]
print('xcd=$x', d['ms_per_step'], 'ms;  3x3-192 w2048:', r['avg_launch_ms'], 'ms', r['frac'], '; dominant frac', d['roofline']['frac'])
PY
done; done
echo done
