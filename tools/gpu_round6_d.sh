#!/bin/bash
# Round 6, D: the weight sweep alone with its full failure text; then the rehearsals call C lost to the process guard
# (six ranks + the launcher were seven GPU processes: four ranks here)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6d
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_weight_sweep.py -q -m gpu > $O/sweep.txt 2>&1; tail -60 $O/sweep.txt | cut -c1-250
cat gpurun_out/weight_sweep_ties.json
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$1', 'MPix/s', d['value'], 'ms/step', d['ms_per_step'], 'n_gpus', d['n_gpus'], 'frames/GPU', c['frames_per_gpu'], 'cores', c['cores_per_rank'], 'busy', c['host_cores_busy'], 'waits', c.get('host_waits'), '|', c['workload'][-40:])"; }
timeout -k 10 300 python bench.py --gpus 4 --share-gpu --frames-per-gpu 2 --steps 3 --warmup 1 --prime 1 --no-cpu-baseline --no-extras 2> $O/share4.err | tee $O/share4.json | line "4 ranks x 2 frames on ONE GPU:" | tee $O/rehearsal.txt
timeout -k 10 300 python bench.py --gpus 2 --share-gpu --frames-per-gpu 4 --steps 3 --warmup 1 --prime 1 --no-cpu-baseline --no-extras 2> $O/share2.err | tee $O/share2.json | line "2 ranks x 4 frames on ONE GPU:" | tee -a $O/rehearsal.txt
for n in 1 8; do
  PCONV_ENGINE_TIMING=1 timeout -k 10 300 python bench.py --emulate-local-world $n --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2> $O/emul_$n.err | line "emulate-local-world $n:" | tee -a $O/rehearsal.txt
  grep "decode 8" $O/emul_$n.err | tail -1 | cut -c1-170 | tee -a $O/rehearsal.txt
done
echo done
