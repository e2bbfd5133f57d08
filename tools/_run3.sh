for t in 0 8 9; do
  echo "== PCONV_CONV_TUNE=$t"
  PCONV_CONV_TUNE=$t python tools/gpu_probe_conv.py 192 192 3 1 64 2048 5 | sed -n 1,2p
  PCONV_CONV_TUNE=$t python tools/gpu_probe_conv.py 192 768 3 1 32 1024 5 | sed -n 1,2p
  PCONV_CONV_TUNE=$t python tools/gpu_probe_conv.py 192 192 3 1 16 512 20 | sed -n 1,2p
  PCONV_CONV_TUNE=$t python tools/gpu_probe_conv.py 96 192 1 1 32 1024 20 | sed -n 1,2p
done
