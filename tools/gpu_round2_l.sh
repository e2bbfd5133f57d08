#!/bin/bash
# session L: parallel group coding in encode; pipelined decode (chunks 0 / 2 / 4)
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_engine.py -m gpu -x -q > gpurun_out/r2l_pytest.log 2>&1 || { tail -40 gpurun_out/r2l_pytest.log; exit 1; }
grep -q "Memory access fault" gpurun_out/r2l_pytest.log && exit 1
tail -3 gpurun_out/r2l_pytest.log
for c in 0 4 2; do
echo "== decode chunk $c"
PCONV_DECODE_CHUNK=$c python bench.py --steps 3 --no-cpu-baseline 2>/dev/null | cut -c1-150
done
echo done
