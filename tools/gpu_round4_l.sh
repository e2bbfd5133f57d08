#!/bin/bash
# timing ablations of the F(4x2,3x3) kernel (wrong results by construction): what each part costs
for v in w42a abl_NOPDMA abl_NOWDMA abl_NOTRANSFORM abl_NOBAR abl_NOWAYOUT abl_ALL w42a; do
  echo "== $v: $(PCONV_HIP_LIB=$PWD/tools/_build/libpconv_hip_$v.so PCONV_PROBE_SHORT=1 PCONV_PROBE_NODIRECT=1 timeout -k 10 120 python tools/gpu_probe_wino42.py 2>&1 | grep "3x3 192->192" | sed 's/wino [0-9.]* ms ([0-9]* TF alg, diff 0)//; s/diff.*//')"
done 2>&1 | tee gpurun_out/r4l_ablation.txt
