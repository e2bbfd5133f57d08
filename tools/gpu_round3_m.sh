#!/bin/bash
# round 3, session M: fat-wave 1x1 / GDN kernel -- parity (bit-exact against the oracle chain and the tiled kernel), timing
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q > gpurun_out/r3m_pytest.log 2>&1 || { tail -50 gpurun_out/r3m_pytest.log; exit 1; }
tail -3 gpurun_out/r3m_pytest.log
: > gpurun_out/r3m_1x1.txt
for v in tiled fat; do
  echo "== PCONV_CONV1X1=$v" >> gpurun_out/r3m_1x1.txt
  PCONV_CONV1X1=$v timeout -k 10 120 python tools/gpu_probe_1x1.py 2>/dev/null >> gpurun_out/r3m_1x1.txt
done
cat gpurun_out/r3m_1x1.txt
