#!/bin/bash
# Round 6, C: the WHOLE GPU suite on the current tree (new: weight sweep, trained model, frame I/O, row split), the
# one-frame A/B of the step kernel's two knobs, then the multi-GPU rehearsals config #5 can take on one box:
# strong-scaling shard of 64 frames on one rank (8 calls of 8), six ranks sharing the GPU (the box allows six GPU
# processes), one rank on 1/8 of the host (sleeping waits = blocking events).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $O/gpu_tests.txt
for cfg in "0 1" "0 2" "1 1" "0 1" "0 2"; do set -- $cfg; PCONV_EE_XCD=$1 PCONV_EE_CONTIG=$2 timeout -k 10 200 python tools/gpu_probe_decode_modes.py 1 2>&1 | grep decode | tail -2 | sed "s/^/xcd=$1 contig=$2 /"; done | tee $O/decode_ab_one_frame.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$1', 'MPix/s', d['value'], 'ms/step', d['ms_per_step'], 'n_gpus', d['n_gpus'], 'frames/GPU', c['frames_per_gpu'], 'cores', c['cores_per_rank'], 'busy', c['host_cores_busy'], 'waits', c.get('host_waits'), 'resident', c.get('value_frames_resident'), 'one_frame', d.get('value_one_frame'), '|', c['workload'][-40:])"; }
timeout -k 10 300 python bench.py --frames-total 64 --gpus 1 --steps 2 --warmup 1 --prime 1 --no-cpu-baseline --no-extras 2> $O/strong64.err | tee $O/strong64.json | line "strong 64 frames on 1 rank:" | tee $O/rehearsal.txt
timeout -k 10 300 python bench.py --gpus 6 --share-gpu --frames-per-gpu 1 --steps 3 --warmup 1 --prime 1 --no-cpu-baseline --no-extras 2> $O/share6.err | tee $O/share6.json | line "6 ranks x 1 frame on ONE GPU:" | tee -a $O/rehearsal.txt
timeout -k 10 300 python bench.py --gpus 2 --share-gpu --frames-per-gpu 4 --steps 3 --warmup 1 --prime 1 --no-cpu-baseline --no-extras 2> $O/share2.err | tee $O/share2.json | line "2 ranks x 4 frames on ONE GPU:" | tee -a $O/rehearsal.txt
for n in 1 8; do
  PCONV_ENGINE_TIMING=1 timeout -k 10 300 python bench.py --emulate-local-world $n --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2> $O/emul_$n.err | line "emulate-local-world $n:" | tee -a $O/rehearsal.txt
  grep "decode 8" $O/emul_$n.err | tail -1 | cut -c1-170 | tee -a $O/rehearsal.txt
done
echo done
