#!/bin/bash
# Round 5, K: step ranges interleaved over the groups of the tail encode, against group by group; number of ranges.
O=$PWD/gpurun_out/r5k
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_entropy_mfma.py tests/test_gpu_codec_vs_oracle.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$1', 'MPix/s', d['value'], 'ms/step', d['ms_per_step'])"; }
for rep in 1 2; do
  for cfg in "0 4" "1 4" "1 3" "1 6" "1 2"; do
    set -- $cfg
    PCONV_ENGINE_ENCODE_INTERLEAVE=$1 PCONV_ENGINE_ENCODE_RANGES=$2 PCONV_ENGINE_TIMING=1 timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check 2> $O/err.txt | line "bench [interleave $1 ranges $2] rep $rep:"
    grep "encode 2" $O/err.txt | tail -1 | cut -c1-150
  done
done 2>&1 | tee $O/bench.txt
for n in 1 2 4; do for il in 0 1; do PCONV_ENGINE_ENCODE_INTERLEAVE=$il python bench.py --frames-per-gpu $n --steps 3 --warmup 1 --no-cpu-baseline --no-check 2>/dev/null | line "frames $n interleave $il:"; done; done | tee $O/bench_frames.txt
