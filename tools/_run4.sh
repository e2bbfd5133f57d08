set -e
echo "== baseline"; python tools/gpu_probe_conv.py 192 192 3 1 64 2048 5 | sed -n 1,1p
for v in 1 2 4 6 7; do
  echo "== CONV_ABL=$v (1 no barrier, 2 no global->LDS staging, 4 no LDS operand reads)"
  PCONV_HIP_LIB=$PWD/tools/_build/libpconv_hip_abl$v.so python tools/gpu_probe_conv.py 192 192 3 1 64 2048 5 | sed -n 1,1p
done
