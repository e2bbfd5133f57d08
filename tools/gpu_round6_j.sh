#!/bin/bash
# Round 6, J: GPU_MAX_HW_QUEUES (HIP runtime knob, read at initialisation; default 4 hardware queues per process) with
# the frame pipe's two copy streams in the process: single rank at 8 / 4 / 2 / 1 frames per call, host to host
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6j
mkdir -p $O
cd $R
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); c=d['config']; e=d.get('entropy',{}); print('$1', 'MPix/s', d['value'], 'ms/step', d['ms_per_step'], 'decode_ms', e.get('decode_ms'), 'encode_ms', e.get('encode_ms'), 'busy', c['host_cores_busy'])"; }
for rep in 1 2; do for q in 4 8; do
GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras 2> $O/e.err | line "1 rank x 8 frames, host to host, GPU_MAX_HW_QUEUES=$q:" | tee -a $O/queues.txt
done; done
for f in 4 2 1; do for q in 4 8; do
GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python bench.py --frames-per-gpu $f --steps 4 --warmup 1 --no-cpu-baseline --no-extras 2> $O/e.err | line "1 rank x $f frames, host to host, GPU_MAX_HW_QUEUES=$q:" | tee -a $O/queues.txt
done; done
for q in 4 8; do
GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python bench.py --frames-per-gpu 4 --steps 4 --warmup 1 --no-cpu-baseline --no-extras --io resident 2> $O/e.err | line "1 rank x 4 frames, resident (no pipe), GPU_MAX_HW_QUEUES=$q:" | tee -a $O/queues.txt
done
echo done
