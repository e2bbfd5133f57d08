#!/bin/bash
# Round 6, E: the weight sweep (fixed: CPU random numbers onto GPU parameters); the shared-GPU rehearsal with the
# frames resident and host to host (call D: 4 x 2 frames 48.6 MPix/s against round 5's 70.1 -- which of the two?)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6e
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_weight_sweep.py -q -m gpu > $O/sweep.txt 2>&1; tail -15 $O/sweep.txt | cut -c1-250
cat gpurun_out/weight_sweep_ties.json
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$1', 'MPix/s', d['value'], 'ms/step', d['ms_per_step'], 'n_gpus', d['n_gpus'], 'frames/GPU', c['frames_per_gpu'], 'cores', c['cores_per_rank'], 'busy', c['host_cores_busy'], '|', c['residency'][:40])"; }
for io in resident host resident host; do
timeout -k 10 300 python bench.py --gpus 4 --share-gpu --frames-per-gpu 2 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --io $io 2> $O/share4_$io.err | line "4 ranks x 2 frames on ONE GPU, io=$io:" | tee -a $O/rehearsal.txt
done
for io in resident host; do
timeout -k 10 300 python bench.py --gpus 2 --share-gpu --frames-per-gpu 4 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --io $io 2> $O/share2_$io.err | line "2 ranks x 4 frames on ONE GPU, io=$io:" | tee -a $O/rehearsal.txt
done
echo done
