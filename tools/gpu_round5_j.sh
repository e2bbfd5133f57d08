#!/bin/bash
# Round 5, J: slice / uslice with 16-byte stores (tests + probe + in-bench rate).
O=$PWD/gpurun_out/r5j
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_codec_vs_oracle.py -x -q -m gpu -k "slice or chain or surface or ops" 2>&1 | tail -3 | tee $O/tests.txt
for rb in 4 2 1; do echo "== PCONV_RESAMPLE_ROWS=$rb"; PCONV_RESAMPLE_ROWS=$rb python tools/gpu_probe_hbm.py 2>&1 | grep -i "slice"; done | tee $O/resample_rows.txt
for rb in 2 1; do PCONV_RESAMPLE_ROWS=$rb python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rows $rb:', d['value'], d['ms_per_step'])
for r in d['hbm']:
    if 'slice' in r['kernel'] or 'clip' in r['kernel']: print('  ', r['kernel'], r['launches'], r['avg_launch_us'], r['achieved'], r['frac'], r.get('traffic'))
"; done | tee $O/bench_hbm.txt
