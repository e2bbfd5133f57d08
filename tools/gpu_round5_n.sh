#!/bin/bash
# Round 5, N: more hardware queues for the decoder's chains (GPU_MAX_HW_QUEUES, a ROCm runtime knob read at HIP
# initialisation: default 4 -- round 3 found that more than four decoder chains wait for each other).
O=$PWD/gpurun_out/r5n
mkdir -p $O
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'MPix/s', d['value'], 'ms/step', d['ms_per_step'], 'busy', d['config']['host_cores_busy'])"; }
for cfg in "4 - -" "8 4 -" "8 8 host" "8 8 queued" "8 6 host" "16 8 host" "16 8 queued" "8 4 queued"; do
  set -- $cfg
  ( export GPU_MAX_HW_QUEUES=$1
    [ "$2" != "-" ] && export PCONV_ENGINE_GROUPS=$2
    [ "$3" != "-" ] && export PCONV_ENGINE_CHAIN=$3
    PCONV_ENGINE_TIMING=1 timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check 2> $O/err.txt | line "hw queues $1 groups $2 chain $3:"
    grep "decode 8" $O/err.txt | tail -1 | cut -c1-150 )
done 2>&1 | tee $O/hw_queues.txt
