set -e
python -m pytest tests/test_gpu_ops.py -q -k conv 2>&1 | tail -1
PCONV_CONV_TUNE=16 python -m pytest tests/test_gpu_ops.py -q -k conv 2>&1 | tail -1
for t in 0 1 16; do
  echo "== PCONV_CONV_TUNE=$t (192-class)"
  PCONV_CONV_TUNE=$t python tools/gpu_probe_conv.py 192 192 3 1 64 2048 5 | sed -n 1,2p
  PCONV_CONV_TUNE=$t python tools/gpu_probe_conv.py 192 768 3 1 32 1024 5 | sed -n 1,2p
  PCONV_CONV_TUNE=$t python tools/gpu_probe_conv.py 192 192 3 1 16 512 20 | sed -n 1,2p
  PCONV_CONV_TUNE=$t python tools/gpu_probe_conv.py 192 192 3 1 8 256 20 | sed -n 1,2p
  PCONV_CONV_TUNE=$t python tools/gpu_probe_conv.py 96 192 1 1 32 1024 20 | sed -n 1,2p
  PCONV_CONV_TUNE=$t python tools/gpu_probe_conv.py 192 192 3 2 32 1024 20 | sed -n 1,2p
done
