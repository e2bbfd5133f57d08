#!/bin/bash
# round 3, session G: Winograd kernel v3 (barrier without vmcnt(0)) + ablations (timing only: wrong results)
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_wino.py -m gpu -x -q > gpurun_out/r3g_pytest.log 2>&1 || { tail -60 gpurun_out/r3g_pytest.log; exit 1; }
tail -3 gpurun_out/r3g_pytest.log
: > gpurun_out/r3g_wino.txt
for v in base WNOEPI WNOTR WNOBOTH; do
  echo "== variant $v" >> gpurun_out/r3g_wino.txt
  if [ $v = base ]; then unset PCONV_HIP_LIB; else export PCONV_HIP_LIB=$PWD/tools/_build/libpconv_hip_$v.so; fi
  timeout -k 10 300 python tools/gpu_probe_wino.py 2>/dev/null | cut -c1-130 >> gpurun_out/r3g_wino.txt
done
unset PCONV_HIP_LIB
cat gpurun_out/r3g_wino.txt
