#!/bin/bash
# session F: queued-ahead decoder chain -- parity tests, then decode timings
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_codec_vs_oracle.py -m gpu -x -q --deselect tests/test_gpu_codec_vs_oracle.py::test_engine_equals_oracle_at_reference_size > gpurun_out/r2f_pytest.log 2>&1 || { tail -40 gpurun_out/r2f_pytest.log; exit 1; }
grep -q "Memory access fault" gpurun_out/r2f_pytest.log && exit 1
tail -3 gpurun_out/r2f_pytest.log
python __graft_entry__.py --smoke 2>&1 | tail -1
PCONV_ENGINE_TIMING=1 timeout -k 10 300 python tools/gpu_probe_engine.py --batch --batch8 > gpurun_out/r2f_probe_engine.log 2>&1 || { tail -5 gpurun_out/r2f_probe_engine.log; exit 1; }
grep "rep1\|decode" gpurun_out/r2f_probe_engine.log
PCONV_ENGINE_CHAIN=host PCONV_ENGINE_TIMING=1 timeout -k 10 300 python tools/gpu_probe_engine.py --batch --batch8 > gpurun_out/r2f_probe_engine_host.log 2>&1 || { tail -5 gpurun_out/r2f_probe_engine_host.log; exit 1; }
echo "== host-driven chain"; grep "rep1" gpurun_out/r2f_probe_engine_host.log
echo done
