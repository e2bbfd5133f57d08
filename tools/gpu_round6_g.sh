#!/bin/bash
# Round 6, G: is it the number of streams?  2 ranks x 4 frames on ONE GPU: resident without the frame pipe (no copy
# streams), host to host with two copy streams and with one; the round-5 tree as the anchor on the same box.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6g
mkdir -p $O
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); c=d['config']; print('$1', 'MPix/s', d['value'], 'ms/step', d['ms_per_step'], 'n_gpus', d['n_gpus'], 'frames/GPU', c['frames_per_gpu'], 'busy', c['host_cores_busy'])"; }
cd $R/_r5tree
timeout -k 10 300 python bench.py --gpus 2 --share-gpu --frames-per-gpu 4 --steps 2 --warmup 1 --no-cpu-baseline 2> $O/r5_2.err | line "round-5 tree, 2 x 4:" | tee -a $O/ab.txt
cd $R
timeout -k 10 300 python bench.py --gpus 2 --share-gpu --frames-per-gpu 4 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --io resident 2> $O/a.err | line "current, 2 x 4, resident, no frame pipe:" | tee -a $O/ab.txt
timeout -k 10 300 python bench.py --gpus 2 --share-gpu --frames-per-gpu 4 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --io host 2> $O/b.err | line "current, 2 x 4, host to host, two copy streams:" | tee -a $O/ab.txt
PCONV_FRAMEPIPE_STREAMS=1 timeout -k 10 300 python bench.py --gpus 2 --share-gpu --frames-per-gpu 4 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --io host 2> $O/c.err | line "current, 2 x 4, host to host, one copy stream:" | tee -a $O/ab.txt
PCONV_WINO_SPLIT=0 timeout -k 10 300 python bench.py --gpus 2 --share-gpu --frames-per-gpu 4 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --io resident 2> $O/d.err | line "current, 2 x 4, resident, no frame pipe, no row split:" | tee -a $O/ab.txt
echo done
