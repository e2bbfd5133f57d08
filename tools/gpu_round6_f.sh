#!/bin/bash
# Round 6, F: the shared-GPU rehearsal, round-5 tree against the current one on the SAME box (call D / E measured
# 2 x 4 frames at 70 MPix/s where round 5's run had 93: the box, or the code?)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6f
mkdir -p $O
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); c=d['config']; print('$1', 'MPix/s', d['value'], 'ms/step', d['ms_per_step'], 'n_gpus', d['n_gpus'], 'frames/GPU', c['frames_per_gpu'], 'busy', c['host_cores_busy'])"; }
for rep in 1 2; do
cd $R/_r5tree
timeout -k 10 300 python bench.py --gpus 2 --share-gpu --frames-per-gpu 4 --steps 2 --warmup 1 --no-cpu-baseline 2> $O/r5_2.err | line "round-5 tree, 2 x 4:" | tee -a $O/ab.txt
cd $R
timeout -k 10 300 python bench.py --gpus 2 --share-gpu --frames-per-gpu 4 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --io resident 2> $O/r6_2.err | line "current tree, 2 x 4, resident:" | tee -a $O/ab.txt
done
cd $R/_r5tree
timeout -k 10 300 python bench.py --gpus 4 --share-gpu --frames-per-gpu 2 --steps 2 --warmup 1 --no-cpu-baseline 2> $O/r5_4.err | line "round-5 tree, 4 x 2:" | tee -a $O/ab.txt
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2> $O/r5_1.err | line "round-5 tree, 1 x 8:" | tee -a $O/ab.txt
cd $R
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --io resident 2> $O/r6_1.err | line "current tree, 1 x 8, resident:" | tee -a $O/ab.txt
echo done
