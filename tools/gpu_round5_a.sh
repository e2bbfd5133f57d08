#!/bin/bash
# Round 5, A: one rank's share of the host.  ONE real rank on one MI355X, pinned to
# quota / N CPUs (N = 1, 2, 4, 8), LOCAL_WORLD_SIZE = N for the engine's host rules; then the engine's
# knobs at the 2- and 4-CPU shares.
O=$PWD/gpurun_out/r5a
mkdir -p $O
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$1', 'MPix/s', d['value'], 'ms/step', d['ms_per_step'], 'cores', c['cores_per_rank'], 'host_cores_busy', c['host_cores_busy'])"; }
cat /sys/fs/cgroup/cpu.max; nproc; python -c "import os; print(len(os.sched_getaffinity(0)))"
for rep in 1 2; do
for n in 1 2 4 8; do
  PCONV_ENGINE_TIMING=1 timeout -k 10 400 python bench.py --emulate-local-world $n --steps 3 --warmup 1 --no-cpu-baseline --no-check 2> $O/err_n$n.txt | line "emulate $n rep $rep:"
  grep "decode 8" $O/err_n$n.txt | tail -1 | cut -c1-160
  grep "encode 2" $O/err_n$n.txt | tail -1 | cut -c1-160
done
done 2>&1 | tee $O/host_share_default.txt
for n in 8 4; do
for cfg in "GROUPS=2" "GROUPS=1" "GROUPS=4 CHAIN=queued" "GROUPS=2 CHAIN=queued" "SPIN_US=0" "SPIN_US=2000"; do
  ( for kv in $cfg; do export PCONV_ENGINE_$kv; done
    PCONV_ENGINE_TIMING=1 timeout -k 10 400 python bench.py --emulate-local-world $n --steps 3 --warmup 1 --no-cpu-baseline --no-check 2> $O/err_k.txt | line "emulate $n [$cfg]:"
    grep "decode 8" $O/err_k.txt | tail -1 | cut -c1-160 )
done
done 2>&1 | tee $O/host_share_knobs.txt
