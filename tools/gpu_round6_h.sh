#!/bin/bash
# Round 6, H: copy streams created through the C ABI instead of torch's stream pool: the shared-GPU rehearsal again
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6h
mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_frame_io.py -x -q -m gpu 2>&1 | tail -2
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); c=d['config']; print('$1', 'MPix/s', d['value'], 'ms/step', d['ms_per_step'], 'n_gpus', d['n_gpus'], 'frames/GPU', c['frames_per_gpu'], 'cores', c['cores_per_rank'], 'busy', c['host_cores_busy'], 'resident', c.get('value_frames_resident'), 'one', d.get('value_one_frame'))"; }
timeout -k 10 300 python bench.py --gpus 2 --share-gpu --frames-per-gpu 4 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2> $O/a.err | line "2 ranks x 4 frames on ONE GPU, host to host:" | tee -a $O/rehearsal.txt
timeout -k 10 300 python bench.py --gpus 4 --share-gpu --frames-per-gpu 2 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2> $O/b.err | line "4 ranks x 2 frames on ONE GPU, host to host:" | tee -a $O/rehearsal.txt
timeout -k 10 300 python bench.py --gpus 5 --share-gpu --frames-per-gpu 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2> $O/c.err | line "5 ranks x 1 frame on ONE GPU, host to host:" | tee -a $O/rehearsal.txt
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2> $O/d.err | tee $O/bench.json | line "1 rank x 8 frames, host to host:" | tee -a $O/rehearsal.txt
echo done
