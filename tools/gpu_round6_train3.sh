#!/bin/bash
# Round 6, training call 3: stage 2 again (call 2 trained it and lost it at the export), export, evaluation through the
# codec; then, with the trained weights: parity against the oracle at 512x1024 and at the metric size, the bench line.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/train_r6
mkdir -p $O $R/trained/r6
cd $R
python tools/train_round6.py stage2 --minutes ${S2_MIN:-6.5} --lr 1e-4 --clip 0.5 --workers 4 --from trained/r6/stage1_b.pack.pt > $O/stage2_stdout.txt 2>&1 || { tail -20 $O/stage2_stdout.txt; exit 1; }
grep "Test set" $O/stage2_stdout.txt | tail -4
cat $O/stage2_report.json
cp $O/codec_3_56.pack.pt $R/trained/r6/codec_3_56.pack.pt
timeout -k 10 900 python -m pytest tests/test_gpu_trained.py -x -q -m gpu 2>&1 | tail -5 | tee $O/trained_tests.txt
cat gpurun_out/trained_parity.json
python tools/weights_pack.py unpack trained/r6/codec_3_56.pack.pt /tmp/trained_r6 > /dev/null
python bench.py --steps 5 --warmup 2 --weights /tmp/trained_r6 --content procedural --no-cpu-baseline > $O/bench_trained.json 2> $O/bench_trained.err || tail -20 $O/bench_trained.err
cut -c1-1800 $O/bench_trained.json
echo done
