#!/bin/bash
# Round 5, M: rehearsal of the N-process path on ONE GPU (bench.py --share-gpu: every rank on cuda:0, gloo reduction).
O=$PWD/gpurun_out/r5m
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_entropy_mfma.py -x -q -m gpu 2>&1 | tail -2
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); c=d['config']; print('$1', 'n_gpus', d['n_gpus'], 'MPix/s', d['value'], 'ms/step', d['ms_per_step'], 'frames/rank', c['frames_per_gpu'], 'cores/rank', c['cores_per_rank'], 'rank0 host_cores_busy', c['host_cores_busy'], c.get('share_gpu',''))"; }
timeout -k 10 400 python bench.py --frames-per-gpu 8 --steps 2 --warmup 1 --no-cpu-baseline 2> $O/err1.txt | line "1 rank x 8 frames:"
timeout -k 10 500 python bench.py --gpus 2 --share-gpu --frames-per-gpu 4 --steps 2 --warmup 1 --no-cpu-baseline 2> $O/err2.txt | line "2 ranks x 4 frames, one GPU:"
timeout -k 10 500 python bench.py --gpus 4 --share-gpu --frames-per-gpu 2 --steps 2 --warmup 1 --no-cpu-baseline 2> $O/err4.txt | line "4 ranks x 2 frames, one GPU:"
tail -3 $O/err4.txt
