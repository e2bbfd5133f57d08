#!/bin/bash
# round 4, call C: where the band kernels' time goes (kernel trace of the entropy engine alone) + the flag probe
set -o pipefail
mkdir -p gpurun_out
O=$PWD/gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
for n in 1 8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ee_$n -- python3 $R/tools/gpu_probe_entropy_only.py $n 3 > $O/r4c_ee_$n.txt 2> $O/r4c_ee_$n.err || { tail -5 $O/r4c_ee_$n.err; exit 1; }
  cat $O/r4c_ee_$n.txt
  f=$(find /tmp/prof_ee_$n -name "*kernel_stats.csv" | head -1)
  cp "$f" $O/r4c_ee_${n}_kernel_stats.csv
  head -12 "$f" | cut -c1-220
done
cd $R
timeout -k 10 300 tools/_build/flag_chain_probe 780 > $O/r4c_flag_probe.txt 2>&1 || { tail -5 $O/r4c_flag_probe.txt; }
cat $O/r4c_flag_probe.txt
