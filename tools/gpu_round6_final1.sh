#!/bin/bash
# Round 6 evidence, part 1: smoke, the whole GPU suite (trained weights present: tests/test_gpu_trained.py runs), the
# headline bench host to host with the CPU baseline at the metric size, the same with the trained model, frames 1/2/4,
# config #3 tables.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final6
mkdir -p $O
cd $R
python __graft_entry__.py --smoke 2>&1 | tail -1
timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee $O/gpu_tests.txt
PCONV_BENCH_TABLE=1 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err || { tail -30 $O/bench_n1.err; exit 1; }
cut -c1-1200 $O/bench_n1.json
python tools/weights_pack.py unpack trained/r6/codec_3_56.pack.pt /tmp/trained_r6 > /dev/null
python bench.py --weights /tmp/trained_r6 --content procedural --cpu-sample 1024x2048 > $O/bench_trained.json 2> $O/bench_trained.err || { tail -30 $O/bench_trained.err; exit 1; }
cut -c1-1200 $O/bench_trained.json
for n in 1 2 4; do python bench.py --frames-per-gpu $n --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | cut -c1-140; done | tee $O/bench_frames_1_2_4.txt
python bench.py --mode analysis --steps 5 --warmup 2 > $O/analysis_1024x2048.json 2> $O/analysis.err || { tail -30 $O/analysis.err; exit 1; }
cut -c1-300 $O/analysis_1024x2048.json
PCONV_BENCH_TABLE=1 python bench.py --mode analysis --height 2048 --width 4096 --steps 3 --warmup 1 > $O/analysis_4096x2048.json 2>> $O/analysis.err || { tail -30 $O/analysis.err; exit 1; }
cut -c1-300 $O/analysis_4096x2048.json
echo done
