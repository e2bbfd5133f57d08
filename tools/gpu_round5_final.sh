#!/bin/bash
# Round 5 evidence: smoke, whole GPU suite, headline bench (with the CPU baseline at the metric size), frames 1/2/4,
# config #3 tables, rocprofv3 kernel stats of the bench command, PMC passes (separate runs, 8 frames per GPU like the
# bench itself) summarised per kernel, the host-share sweep.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final5
mkdir -p $O
cd $R
python __graft_entry__.py --smoke 2>&1 | tail -1
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee $O/gpu_tests.txt
PCONV_BENCH_TABLE=1 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err || { tail -30 $O/bench_n1.err; exit 1; }
cut -c1-900 $O/bench_n1.json
python bench.py --mode analysis --steps 5 --warmup 2 > $O/analysis_1024x2048.json 2> $O/analysis.err || { tail -30 $O/analysis.err; exit 1; }
cut -c1-300 $O/analysis_1024x2048.json
PCONV_BENCH_TABLE=1 python bench.py --mode analysis --height 2048 --width 4096 --steps 3 --warmup 1 > $O/analysis_4096x2048.json 2>> $O/analysis.err || { tail -30 $O/analysis.err; exit 1; }
cut -c1-300 $O/analysis_4096x2048.json
for n in 1 2 4; do python bench.py --frames-per-gpu $n --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | cut -c1-140; done | tee $O/bench_frames_1_2_4.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$1', 'MPix/s', d['value'], 'ms/step', d['ms_per_step'], 'cores', c['cores_per_rank'], 'host_cores_busy', c['host_cores_busy'])"; }
for n in 1 2 4 8; do
  PCONV_ENGINE_TIMING=1 timeout -k 10 400 python bench.py --emulate-local-world $n --steps 3 --warmup 1 --no-cpu-baseline --no-check 2> $O/err_n$n.txt | line "emulate $n:"
  grep "decode 8" $O/err_n$n.txt | tail -1 | cut -c1-160
done 2>&1 | tee $O/host_share_final.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 $R/bench.py --steps 1 --warmup 0 --prime 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err || tail -5 $O/bench_under_rocprof.err
cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
python3 $R/tools/gpu_idle_map.py $(find /tmp/prof_bench -name "*kernel_trace.csv" | head -1) 12 50 > $O/bench_idle_map.txt 2>&1 || true
head -8 $O/bench_idle_map.txt
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1)); rm -rf /tmp/pmc_b_$i
  PCONV_BENCH_TABLE=1 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_b_$i -- python3 $R/bench.py --steps 1 --warmup 0 --prime 1 --no-cpu-baseline --no-check > $O/pmc_pass_$i.json 2> $O/pmc_pass_$i.err || { tail -5 $O/pmc_pass_$i.err; }
done
python3 $R/tools/summarise_pmc.py $O/bench_pmc.json /tmp/pmc_b_1 /tmp/pmc_b_2 /tmp/pmc_b_3 /tmp/pmc_b_4 /tmp/pmc_b_5 /tmp/pmc_b_6 --bench-json $O/pmc_pass_1.json
echo done
