#!/bin/bash
# Round 5, B: (1) the matrix-core encoder kernel: identical streams + timing against the vector kernel, its GPU tests;
# (2) host-share knobs at 2 and 4 CPUs per rank: thread confinement, serial decoding per group, blocking waits.
O=$PWD/gpurun_out/r5b
mkdir -p $O
{
python tools/gpu_probe_entropy_mfma.py 1 3 2 64 && python tools/gpu_probe_entropy_mfma.py 1 3 4 128 && \
python tools/gpu_probe_entropy_mfma.py 1 3 16 512 && python tools/gpu_probe_entropy_mfma.py 2 3 16 512 && python tools/gpu_probe_entropy_mfma.py 8 3 16 512
PCONV_EE_MFMA_WAVES=8 python tools/gpu_probe_entropy_mfma.py 1 3 16 512 && PCONV_EE_MFMA_WAVES=8 python tools/gpu_probe_entropy_mfma.py 8 3 16 512
} 2>&1 | tee $O/mfma_probe.txt
timeout -k 10 600 python -m pytest tests/test_gpu_entropy_mfma.py tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -5 | tee $O/mfma_tests.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$1', 'MPix/s', d['value'], 'ms/step', d['ms_per_step'], 'cores', c['cores_per_rank'], 'host_cores_busy', c['host_cores_busy'])"; }
for n in 8 4; do
for cfg in "X=0" "SPIN_US=0" "WORKERS=1" "WORKERS=1 BLOCKING_SYNC=1" "BLOCKING_SYNC=1" "GROUPS=2 CHAIN=queued" "GROUPS=2 WORKERS=1"; do
  ( for kv in $cfg; do export PCONV_ENGINE_$kv; done
    PCONV_BENCH_THREADS=1 PCONV_ENGINE_TIMING=1 timeout -k 10 400 python bench.py --emulate-local-world $n --steps 3 --warmup 1 --no-cpu-baseline --no-check 2> $O/err_k.txt | line "emulate $n [$cfg]:"
    grep "decode 8" $O/err_k.txt | tail -1 | cut -c1-160
    grep "bench threads" $O/err_k.txt | head -8 )
done
done 2>&1 | tee $O/host_share_knobs2.txt
