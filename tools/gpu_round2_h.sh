#!/bin/bash
# session H: tables8 kernel parity + kernel trace of one-frame decode (durations and gaps)
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_codec_vs_oracle.py -m gpu -x -q --deselect tests/test_gpu_codec_vs_oracle.py::test_engine_equals_oracle_at_reference_size > gpurun_out/r2h_pytest.log 2>&1 || { tail -40 gpurun_out/r2h_pytest.log; exit 1; }
grep -q "Memory access fault" gpurun_out/r2h_pytest.log && exit 1
tail -3 gpurun_out/r2h_pytest.log
PCONV_ENGINE_TIMING=1 timeout -k 10 300 python tools/gpu_probe_engine.py --batch --batch8 > gpurun_out/r2h_probe_engine.log 2>&1 || { tail -5 gpurun_out/r2h_probe_engine.log; exit 1; }
grep "rep1" gpurun_out/r2h_probe_engine.log
cd /tmp && export TMPDIR=/tmp
for n in 1 4; do
rm -rf /tmp/trace_ee_$n
PCONV_ENGINE_GROUPS=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_ee_$n -- python3 $R/tools/gpu_probe_entropy.py --n=$n --once > $R/gpurun_out/r2h_trace_$n.log 2>&1 || tail -3 $R/gpurun_out/r2h_trace_$n.log
python3 - <<PY
import csv, glob, collections
f = glob.glob('/tmp/trace_ee_$n/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# decode part: from the first ee_scatter_kernel on
names = [r['Kernel_Name'] for r in rows]
first = next(i for i, nme in enumerate(names) if 'ee_scatter' in nme)
dec = rows[first:]
dur = collections.defaultdict(list); gaps = []
for a, b in zip(dec, dec[1:]):
    gaps.append(int(b['Start_Timestamp']) - int(a['End_Timestamp']))
for r in dec:
    k = r['Kernel_Name'].split('(')[0].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    dur[k[:40]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
print('N=$n decode kernels:', len(dec), 'span ms', (int(dec[-1]['End_Timestamp']) - int(dec[0]['Start_Timestamp'])) / 1e6)
for k, v in dur.items(): print('  %-42s n=%6d avg %.2f us  total %.1f ms' % (k, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
g = sorted(gaps); print('  gaps: avg %.2f us median %.2f p90 %.2f total %.1f ms' % (sum(g) / len(g) / 1e3, g[len(g) // 2] / 1e3, g[int(len(g) * .9)] / 1e3, sum(g) / 1e6))
small = [x for x in g if x < 20000]; print('  gaps < 20 us: n=%d avg %.2f us total %.1f ms' % (len(small), sum(small) / len(small) / 1e3, sum(small) / 1e6))
PY
done
echo done
