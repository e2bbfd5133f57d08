#!/bin/bash
# Round 5, H: 16-byte packed CDF rows across PCIe (encoder D2H, decoder zero-copy) against the int32 rows.
O=$PWD/gpurun_out/r5h
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_entropy_mfma.py tests/test_gpu_codec_vs_oracle.py -x -q -m gpu 2>&1 | tail -4 | tee $O/tests.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$1', 'MPix/s', d['value'], 'ms/step', d['ms_per_step'], 'host_cores_busy', c['host_cores_busy'])"; }
for rep in 1 2 3; do
  for cfg in "int32" "packed"; do
    PCONV_ENGINE_ROWS=$cfg PCONV_ENGINE_TIMING=1 timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check 2> $O/err.txt | line "bench [rows $cfg] rep $rep:"
    grep "decode 8" $O/err.txt | tail -1 | cut -c1-150
    grep "encode 2" $O/err.txt | tail -1 | cut -c1-150
  done
done 2>&1 | tee $O/bench.txt
for n in 1 2; do for cfg in int32 packed; do PCONV_ENGINE_ROWS=$cfg python bench.py --frames-per-gpu $n --steps 3 --warmup 1 --no-cpu-baseline --no-check 2>/dev/null | line "frames $n [rows $cfg]:"; done; done | tee $O/bench_frames.txt
