#!/bin/bash
# Round 6, B: the new tests first (frame I/O, weight sweep, row split, odd widths, blocking events), then the decoder's
# workgroup order A/B (PCONV_EE_XCD) with PMC evidence for ee_step_kernel, then the bench line.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6b
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_frame_io.py tests/test_gpu_wino42.py tests/test_gpu_entropy_mfma.py tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -5 | tee $O/tests_new.txt
echo "--- decode A/B: PCONV_EE_XCD (workgroup order) x PCONV_EE_CONTIG (1 = positions e, e + 4 per loop body; 2 = neighbours e, e + 1)" | tee $O/decode_ab.txt
for rep in 1 2; do for cfg in "0 1" "1 1" "0 2" "1 2"; do
  set -- $cfg
  PCONV_EE_XCD=$1 PCONV_EE_CONTIG=$2 timeout -k 10 200 python tools/gpu_probe_decode_modes.py 8 2>&1 | grep decode | tail -2 | sed "s/^/xcd=$1 contig=$2 /" | tee -a $O/decode_ab.txt
done; done
for cfg in "0 1" "1 2"; do set -- $cfg; PCONV_EE_XCD=$1 PCONV_EE_CONTIG=$2 timeout -k 10 200 python tools/gpu_probe_decode_modes.py 1 2>&1 | grep decode | tail -2 | sed "s/^/xcd=$1 contig=$2 /" | tee -a $O/decode_ab.txt; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TC[PC]_[A-Za-z0-9_]*" | sort -u > $O/counters_tcp_tcc.txt
wc -l $O/counters_tcp_tcc.txt
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  for x in 0 1; do
    rm -rf /tmp/pmc_${i}_$x
    PCONV_EE_XCD=$x PCONV_EE_CONTIG=$((x+1)) timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_${i}_$x -- python3 $R/tools/gpu_probe_decode_pmc.py 1 > $O/pmc_${i}_$x.out 2> $O/pmc_${i}_$x.err || { echo "pass $i xcd=$x failed"; tail -3 $O/pmc_${i}_$x.err; }
  done
done
python3 - <<'PY' | tee $O/step_kernel_pmc.txt
import csv, glob, os, collections
for x in (0, 1):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sorted(glob.glob("/tmp/pmc_*_%d" % x)):
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                k = r["Kernel_Name"]
                if "ee_step_kernel" not in k:
                    continue
                k = "ee_step_kernel<" + k.split("ee_step_kernel<")[1].split(">")[0] + ">"
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("PCONV_EE_XCD=%d PCONV_EE_CONTIG=%d  (one lock-step group of two 4096x2048 frames, per launch averages)" % (x, x + 1))
    for k, c in sorted(acc.items()):
        print("  %s" % k)
        for name, v in sorted(c.items()):
            print("    %-34s launches %6d  avg %14.1f" % (name, len(v), sum(v) / len(v)))
PY
cd $R
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err || tail -20 $O/bench.err
cut -c1-1500 $O/bench.json

python tools/weights_pack.py unpack trained/r6/codec_3_56.pack.pt /tmp/trained_r6 > /dev/null && python bench.py --steps 5 --warmup 2 --weights /tmp/trained_r6 --content procedural --no-cpu-baseline > $O/bench_trained.json 2> $O/bench_trained.err || tail -20 $O/bench_trained.err
cut -c1-1500 $O/bench_trained.json
echo done
