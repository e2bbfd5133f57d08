#!/bin/bash
# Round 6, K: the decoder's last layer + table kernel as ONE launch (PCONV_EE_FUSE_TABLES=1): parity (engine tests,
# the oracle comparisons at 512x1024 and at the metric size run on it), then decode time against the two-launch form.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6k
mkdir -p $O
cd $R
PCONV_EE_FUSE_TABLES=1 timeout -k 10 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_codec_vs_oracle.py -x -q -m gpu 2>&1 | tail -6 | tee $O/tests_fused.txt
echo "--- decode, PCONV_EE_FUSE_TABLES 0 / 1 (tools/gpu_probe_decode_modes.py N: N frames, 3 decodes each)" | tee $O/fused_ab.txt
for rep in 1 2; do for f in 0 1; do
  PCONV_EE_FUSE_TABLES=$f timeout -k 10 200 python tools/gpu_probe_decode_modes.py 8 2>&1 | grep decode | tail -2 | sed "s/^/fuse=$f /" | tee -a $O/fused_ab.txt
done; done
for f in 0 1 0 1; do PCONV_EE_FUSE_TABLES=$f timeout -k 10 200 python tools/gpu_probe_decode_modes.py 1 2>&1 | grep decode | tail -2 | sed "s/^/fuse=$f /" | tee -a $O/fused_ab.txt; done
for f in 0 1; do PCONV_EE_FUSE_TABLES=$f timeout -k 10 200 python tools/gpu_probe_decode_modes.py 4 2>&1 | grep decode | tail -2 | sed "s/^/fuse=$f /" | tee -a $O/fused_ab.txt; done
for p in 2 8; do PCONV_EE_FUSE_PPW=$p PCONV_EE_FUSE_TABLES=1 timeout -k 10 200 python tools/gpu_probe_decode_modes.py 8 2>&1 | grep decode | tail -2 | sed "s/^/fuse=1 ppw=$p /" | tee -a $O/fused_ab.txt; done
for p in 2 8; do PCONV_EE_FUSE_PPW=$p PCONV_EE_FUSE_TABLES=1 timeout -k 10 200 python tools/gpu_probe_decode_modes.py 1 2>&1 | grep decode | tail -2 | sed "s/^/fuse=1 ppw=$p /" | tee -a $O/fused_ab.txt; done
echo done
