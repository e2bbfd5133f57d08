#!/bin/bash
# Round 5, V: the decoder's step kernel with its first gathers requested before the slab barrier -- parity tests,
# then the codec with the previous kernel (tools/_build/libpconv_hip_prevstep.so) and the new one, alternating
set -e
O=$PWD/gpurun_out/r5v
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_codec_vs_oracle.py -x -q 2>&1 | tail -4 | tee $O/tests.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'MPix/s', d['value'], 'ms/step', d['ms_per_step'])"; }
for rep in 1 2; do
  for v in prev new; do
    ( [ $v = prev ] && export PCONV_HIP_LIB=$PWD/tools/_build/libpconv_hip_prevstep.so
      PCONV_ENGINE_TIMING=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check 2> $O/err.txt | line "bench [$v] rep $rep:" | tee -a $O/bench.txt
      grep "decode 8" $O/err.txt | tail -1 | cut -c1-160 | tee -a $O/bench.txt )
  done
done
