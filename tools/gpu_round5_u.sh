#!/bin/bash
# Round 5, U: the chip in two partitions (CU masks): the entropy chains on few CUs, the transforms on the rest --
# what each loses alone (the price of running them side by side)
O=$PWD/gpurun_out/r5u
mkdir -p $O
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'MPix/s', d['value'], 'ms/step', d['ms_per_step'])"; }
run() {  # label, engine mask, bench mask
  ( [ "$2" != "-" ] && export PCONV_ENGINE_CU_MASK=$2
    [ "$3" != "-" ] && export PCONV_BENCH_CU_MASK=$3
    PCONV_ENGINE_TIMING=1 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check 2> $O/err.txt | line "$1" | tee -a $O/masks.txt
    grep "decode 8" $O/err.txt | tail -1 | cut -c1-160 | tee -a $O/masks.txt
    grep "encode 2" $O/err.txt | tail -1 | cut -c1-160 | tee -a $O/masks.txt )
}
run "whole chip:" - -
run "chains on 32 CUs:" 0:32 -
run "chains on 64 CUs:" 0:64 -
run "chains on 16 CUs:" 0:16 -
run "transforms on 224 CUs:" - 32:224
run "transforms on 192 CUs:" - 64:192
run "chains 32 / transforms 224 (serial):" 0:32 32:224
exit 0
