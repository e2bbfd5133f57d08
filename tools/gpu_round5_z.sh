#!/bin/bash
# Round 5, Z: the ring-pad kernel's launches by shape (kernel trace of one bench step, grid sizes kept)
O=$PWD/gpurun_out/r5z
mkdir -p $O
export TMPDIR=/tmp
( cd /tmp && ${RINGLIB:+PCONV_HIP_LIB=$RINGLIB} timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/r5z -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --prime 1 --no-cpu-baseline --no-check > $O/bench${TAG}.json 2> $O/bench${TAG}.err )
f=$(find /tmp/r5z -name "p_kernel_trace.csv" | head -1)
python3 - $f <<'PY' | tee $O/ring_classes${TAG}.txt
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'pseudo_pad_ring' in r['Kernel_Name'] or 'pseudo_pad_kernel' in r['Kernel_Name']:
        k = (r['Kernel_Name'].split('(')[0][-24:], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])
        acc[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
tot = 0
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    tot += sum(v)
    print('%-26s grid %8s x %4s x %3s  launches %4d  avg %7.1f us  total %7.2f ms' % (k[0], k[1], k[2], k[3], len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
print('total %.2f ms over prime + 1 step' % (tot / 1e6))
PY
rm -rf /tmp/r5z
exit 0
