for p in 2 3 4 6 8; do echo "== PPW $p"; PCONV_EE_PPW=$p python tools/gpu_probe_engine.py --batch --batch8 2>&1 | grep -v rep0 | grep 2048 | sed 's/analysis.*ent-enc/ent-enc/; s/synthesis.*//'; done
