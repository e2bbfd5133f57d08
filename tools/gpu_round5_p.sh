#!/bin/bash
# Round 5, P: where the four-block encoder kernel's time goes (timing ablations, builds of tools/build_variant.sh)
O=$PWD/gpurun_out/r5p
mkdir -p $O
export TMPDIR=/tmp
for v in ${VARIANTS:-base ee4abl1 ee4abl2 ee4abl4 ee4abl6}; do
  lib=$GRAFT_REPO_ROOT/tools/_build/libpconv_hip_$v.so
  [ $v = base ] && lib=$GRAFT_REPO_ROOT/pseudocylindrical_convolution_amd/libpconv_hip.so
  ( cd /tmp && PCONV_HIP_LIB=$lib PCONV_ENGINE_ENCODE_RANGES=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o p -- python3 $GRAFT_REPO_ROOT/tools/gpu_probe_entropy_mfma.py 1 2 > $O/prof_$v.log 2>&1 )
  f=$(ls $O/prof_$v/*/p_kernel_stats.csv $O/prof_$v/p_kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && grep -E "ee_conv_bulk_mfma4" $f < /dev/null | sed -e 's/(anonymous namespace):://' -e 's/(EeGeom[^"]*"/"/' | cut -d, -f1-4,6 | sed "s/^/$v /" | tee -a $O/kernels.txt
done
exit 0
