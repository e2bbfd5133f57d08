#!/bin/bash
# round 3, session Y: Winograd kernel with the input transform inside the matrix block: tests, timing against
# the committed kernels, in-kernel phase stamps
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_wino.py -m gpu -x -q > gpurun_out/r3y_pytest.log 2>&1 || { tail -40 gpurun_out/r3y_pytest.log; exit 1; }
tail -2 gpurun_out/r3y_pytest.log
VARIANTS="${VARIANTS:-v8 new v8 new}" bash tools/gpu_round3_x.sh > /dev/null
cp gpurun_out/r3x_ablation.txt gpurun_out/r3y_timing.txt
cat gpurun_out/r3y_timing.txt
: > gpurun_out/r3y_stamps.txt
for lib in ${STAMPS:-stamp}; do
  echo "== $lib" >> gpurun_out/r3y_stamps.txt
  PCONV_HIP_LIB=$PWD/tools/_build/libpconv_hip_$lib.so timeout -k 10 200 python tools/gpu_probe_wino_stamps.py >> gpurun_out/r3y_stamps.txt 2>/dev/null
done
cat gpurun_out/r3y_stamps.txt
