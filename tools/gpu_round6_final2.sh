#!/bin/bash
# Round 6 evidence, part 2: the N-process rehearsals on one GPU and the host-share sweep (final tree), rocprofv3 kernel
# stats + idle map of the bench command, PMC passes (separate runs, 8 frames per GPU like the bench) summarised per kernel.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final6
mkdir -p $O
cd $R
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); c=d['config']; print('$1', 'MPix/s', d['value'], 'ms/step', d['ms_per_step'], 'n_gpus', d['n_gpus'], 'frames/GPU', c['frames_per_gpu'], 'cores', c['cores_per_rank'], 'busy', c['host_cores_busy'], 'waits:', c.get('host_waits'))"; }
timeout -k 10 300 python bench.py --gpus 2 --share-gpu --frames-per-gpu 4 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2> $O/share2.err | line "2 ranks x 4 frames on ONE GPU:" | tee $O/rehearsal_final.txt
timeout -k 10 300 python bench.py --gpus 4 --share-gpu --frames-per-gpu 2 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2> $O/share4.err | line "4 ranks x 2 frames on ONE GPU:" | tee -a $O/rehearsal_final.txt
timeout -k 10 300 python bench.py --frames-total 64 --gpus 1 --steps 2 --warmup 1 --prime 1 --no-cpu-baseline --no-extras 2> $O/strong64.err | line "64 frames on one rank (8 calls of 8):" | tee -a $O/rehearsal_final.txt
for n in 1 2 4 8; do
  timeout -k 10 400 python bench.py --emulate-local-world $n --steps 3 --warmup 1 --no-cpu-baseline --no-check --no-extras 2> $O/err_n$n.txt | line "emulate $n:" | tee -a $O/rehearsal_final.txt
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 $R/bench.py --steps 1 --warmup 0 --prime 1 --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err || tail -5 $O/bench_under_rocprof.err
cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
python3 $R/tools/gpu_idle_map.py $(find /tmp/prof_bench -name "*kernel_trace.csv" | head -1) 12 50 > $O/bench_idle_map.txt 2>&1 || true
head -8 $O/bench_idle_map.txt
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1)); rm -rf /tmp/pmc_b_$i
  PCONV_BENCH_TABLE=1 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_b_$i -- python3 $R/bench.py --steps 1 --warmup 0 --prime 1 --no-cpu-baseline --no-check --no-extras > $O/pmc_pass_$i.json 2> $O/pmc_pass_$i.err || { tail -5 $O/pmc_pass_$i.err; }
done
python3 $R/tools/summarise_pmc.py $O/bench_pmc.json /tmp/pmc_b_1 /tmp/pmc_b_2 /tmp/pmc_b_3 /tmp/pmc_b_4 /tmp/pmc_b_5 --bench-json $O/pmc_pass_1.json

cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_like.json 2> $O/bench_driver_like.err; cut -c1-200 $O/bench_driver_like.json
echo done
