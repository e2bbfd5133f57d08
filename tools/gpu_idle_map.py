"""GPU busy / idle map of one bench step from a rocprofv3 --kernel-trace CSV:
   python tools/gpu_idle_map.py <kernel_trace.csv> [n_largest_gaps]
Union of the kernel intervals (all streams) over the span of the trace, the largest gaps with the kernels on
either side, and the busy time per phase (phases split at gaps > 2 ms)."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy, cur_s, cur_e, gaps = 0, rows[0][0], rows[0][1], []
last_name = rows[0][2]
for s, e, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e - t0, last_name, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e >= cur_e:
        last_name = n
busy += cur_e - cur_s
span = t1 - t0
print("span %.1f ms, GPU busy (union over streams) %.1f ms = %.1f %%, %d kernels, %d gaps" % (span / 1e6, busy / 1e6, 100.0 * busy / span, len(rows), len(gaps)))
hist = [0, 0, 0, 0, 0]
tot = [0, 0, 0, 0, 0]
for g in gaps:
    k = 0 if g[0] < 5e3 else 1 if g[0] < 2e4 else 2 if g[0] < 1e5 else 3 if g[0] < 1e6 else 4
    hist[k] += 1; tot[k] += g[0]
for lab, h, t in zip(("<5us", "5-20us", "20-100us", "0.1-1ms", ">1ms"), hist, tot):
    print("  gaps %-9s %6d  total %.2f ms" % (lab, h, t / 1e6))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 15
for g in sorted(gaps, reverse=True)[:n]:
    print("  gap %.3f ms at +%.1f ms  after %s  before %s" % (g[0] / 1e6, g[1] / 1e6, g[2], g[3]))

# coarse timeline: busy fraction per bin and the kernel with the most time in it
if len(sys.argv) > 3:
    binw = float(sys.argv[3]) * 1e6
    nb = int(span / binw) + 1
    b_busy = [0.0] * nb
    b_top = [dict() for _ in range(nb)]
    for s, e, n in rows:
        i0, i1 = int((s - t0) / binw), int((e - t0) / binw)
        for i in range(i0, i1 + 1):
            lo, hi = max(s, t0 + i * binw), min(e, t0 + (i + 1) * binw)
            if hi > lo:
                b_busy[i] += hi - lo
                b_top[i][n] = b_top[i].get(n, 0) + hi - lo
    for i in range(nb):
        if b_busy[i] > 0:
            top = max(b_top[i].items(), key=lambda kv: kv[1])[0]
            print("  +%7.0f ms  kernel time %5.1f %% of bin  %s" % (i * binw / 1e6, 100.0 * b_busy[i] / binw, top[:50]))
