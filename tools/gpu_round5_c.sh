#!/bin/bash
# Round 5, C: matrix-core encoder kernel with the DMA patch prologue (probe + rocprofv3 kernel stats of both forms),
# the whole GPU suite, the headline bench, the host-share sweep with the adaptive host plan.
O=$PWD/gpurun_out/r5c
mkdir -p $O
R=$PWD
{
python tools/gpu_probe_entropy_mfma.py 1 3 16 512 && python tools/gpu_probe_entropy_mfma.py 8 3 16 512
PCONV_EE_MFMA_WAVES=8 python tools/gpu_probe_entropy_mfma.py 8 3 16 512
} 2>&1 | grep -v amdgpu.ids | tee $O/mfma_probe.txt
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_mfma && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mfma -- python3 $R/tools/gpu_probe_entropy_mfma.py 8 2 16 512 > $O/prof_mfma.log 2>&1; cp $(find /tmp/prof_mfma -name "*kernel_stats.csv" | head -1) $O/mfma_probe_kernel_stats.csv; head -8 $O/mfma_probe_kernel_stats.csv | cut -c1-200 )
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee $O/gpu_tests.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$1', 'MPix/s', d['value'], 'ms/step', d['ms_per_step'], 'cores', c['cores_per_rank'], 'host_cores_busy', c['host_cores_busy'])"; }
for rep in 1 2; do
for n in 1 2 4 8; do
  PCONV_BENCH_THREADS=1 PCONV_ENGINE_TIMING=1 timeout -k 10 400 python bench.py --emulate-local-world $n --steps 3 --warmup 1 --no-cpu-baseline --no-check 2> $O/err_n$n.txt | line "emulate $n rep $rep:"
  grep "decode 8" $O/err_n$n.txt | tail -1 | cut -c1-160
  grep "bench threads" $O/err_n$n.txt | head -3
done
done 2>&1 | tee $O/host_share_adaptive.txt
PCONV_EE_BULK=valu timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check 2>/dev/null | line "bulk valu:" | tee $O/bench_valu.txt
