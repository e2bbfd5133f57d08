#!/bin/bash
# Evidence for the ablation figures of DESIGN.md (sections 4 and 5): the tile convolution with parts
# removed (timing only: results are wrong) and the decoder's tuning knobs.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/round2_ablation.txt
{
  echo "# 3x3 192->192 tile convolution, 16 x 64 x 2048 px, executed TFLOP/s (tools/gpu_probe_conv.py)"
  for v in base NOWDMA NOXDMA NODMA NOEPI NOBAR; do
    echo "== $v"
    if [ $v = base ]; then unset PCONV_HIP_LIB; else export PCONV_HIP_LIB=$PWD/tools/_build/libpconv_hip_$v.so; fi
    timeout -k 10 120 python tools/gpu_probe_conv.py 2>&1 | grep "all columns" | tail -2
  done
  unset PCONV_HIP_LIB
  echo "# epilogue variants (tools/gpu_probe_epilogue.py)"
  timeout -k 10 120 python tools/gpu_probe_epilogue.py 2>&1 | grep -v amdgpu | tail -4
  echo "# entropy decode, seconds for N frames at 4096x2048 (tools/gpu_probe_engine.py), knob = value"
  for cfg in "default" "PCONV_EE_JOINT=1" "PCONV_EE_CONTIG=0" "PCONV_EE_PPW=4" "PCONV_ENGINE_GROUPS=4" "PCONV_ENGINE_GROUPS=1" "PCONV_ENGINE_CHAIN=host"; do
    echo "== $cfg"
    if [ "$cfg" = default ]; then
      timeout -k 10 300 python tools/gpu_probe_engine.py --batch --batch8 2>&1 | grep rep1 | grep 2048x4096 | cut -c1-100
    else
      env $cfg timeout -k 10 300 python tools/gpu_probe_engine.py --batch --batch8 2>&1 | grep rep1 | grep 2048x4096 | cut -c1-100
    fi
  done
} > $out 2>&1
cat $out
