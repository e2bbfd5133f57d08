#!/bin/bash
# Round 5, O: the four-block form of the encoder's matrix-core kernel (v_mfma_f32_16x16x1_4b_f32, the default since;
# PCONV_EE_MFMA_FORM=16x4 is the earlier form) -- parity, whole-launch kernel time per form, the codec with each
set -e
O=$PWD/gpurun_out/r5o
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_entropy_mfma.py -x -q 2>&1 | tail -5 | tee $O/tests.txt
export TMPDIR=/tmp
for waves in 4 8; do
  ( cd /tmp && PCONV_EE_MFMA_WAVES=$waves PCONV_ENGINE_ENCODE_RANGES=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$waves -o p -- python3 $GRAFT_REPO_ROOT/tools/gpu_probe_entropy_mfma.py 1 3 > $O/prof$waves.log 2>&1 )
  f=$(ls $O/prof$waves/*/p_kernel_stats.csv $O/prof$waves/p_kernel_stats.csv 2>/dev/null | head -1)
  echo "whole launches (1 frame x 3 sets), waves per workgroup $waves:" | tee -a $O/kernels.txt
  [ -n "$f" ] && grep -E "ee_conv_bulk" $f < /dev/null | sed -e 's/(anonymous namespace):://' -e 's/(EeGeom[^"]*"/"/' | cut -d, -f1-4 | tee -a $O/kernels.txt
done
[ "$1" = "kernel" ] && exit 0
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'MPix/s', d['value'], 'ms/step', d['ms_per_step'])"; }
for rep in 1 2; do
  for form in 16x4 4b; do
    PCONV_EE_MFMA_FORM=$form PCONV_ENGINE_TIMING=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check 2> $O/err.txt | line "bench [form $form] rep $rep:" | tee -a $O/bench.txt
  done
done
