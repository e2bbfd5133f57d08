#!/bin/bash
# Round 5, W: which library kernels (copies, fills, element-wise) are inside a bench step: kernel stats of 1 and 3 steps
O=$PWD/gpurun_out/r5w
mkdir -p $O
export TMPDIR=/tmp
for k in 1 3; do
  ( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$k -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps $k --warmup 0 --prime 1 --no-cpu-baseline --no-check > $O/s$k.json 2> $O/s$k.err )
  rm -f $O/s$k/p_kernel_trace.csv $O/s$k/*/p_kernel_trace.csv
done
exit 0
