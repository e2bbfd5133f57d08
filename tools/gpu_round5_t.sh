#!/bin/bash
# Round 5, T: the decoder's step kernel alone and beside other chains (kernel trace of the bench at 2 / 8 frames)
O=$PWD/gpurun_out/r5t
mkdir -p $O
export TMPDIR=/tmp
run() {  # name frames groups
  ( cd /tmp && PCONV_ENGINE_GROUPS=$3 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$1 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --frames-per-gpu $2 --steps 1 --warmup 0 --prime 1 --no-cpu-baseline --no-check > $O/$1.json 2> $O/$1.err )
  f=$(ls $O/$1/*/p_kernel_stats.csv $O/$1/p_kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && grep -E "ee_step_kernel|ee_scatter|ee_tables8" $f < /dev/null | sed -e 's/(anonymous namespace):://' -e 's/(EeGeom[^"]*"/"/' | cut -d, -f1-4 | sed "s/^/$1 /" | tee -a $O/kernels.txt
}
run f2g1 2 1
run f2g2 2 2
run f4g2 4 2
run f8g4 8 4
exit 0
