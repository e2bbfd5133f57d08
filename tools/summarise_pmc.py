"""Per-kernel summary of rocprofv3 --pmc passes (counter_collection CSVs) -> JSON.

    python tools/summarise_pmc.py OUT.json DIR [DIR ...] [--trace STATS_DIR] [--flops KERNEL=FLOPS_PER_LAUNCH ...]

Every DIR holds one pass (`rocprofv3 --pmc <counters> --output-format csv -d DIR -- python3 bench.py ...`;
separate passes as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit
one).  Per kernel (template instantiation, argument list stripped): launches and the
average per launch of every counter found.  Corrections of the guide's HBM section:
FETCH_SIZE (KiB) tallies 128-B requests at 64 B on gfx950 -> read bytes = 2 x FETCH_SIZE;
WRITE_SIZE is exact.  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x CUs-share):
reported as the ratio to GRBM_GUI_ACTIVE-free SQ_BUSY_CYCLES per SIMD (see `mfma_busy_note`).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void\s+", "", name)
    name = name.replace("(anonymous namespace)::", "")
    i = name.find("(")
    name = (name[:i] if i > 0 else name).strip()
    # the Winograd kernel's instantiations (residual / plain / Dtow way out) are one kernel to bench.py
    if name.startswith("wino_conv3x3_kernel<"):
        name = "wino_conv3x3_kernel"
    if name.startswith("wino42_conv3x3_kernel<"):
        name = "wino42_conv3x3_kernel"
    return name


def read_pass(d):
    rows = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> values per dispatch
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                k = short(r.get("Kernel_Name", ""))
                rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return rows


def main():
    args = sys.argv[1:]
    out_path, dirs, flops, trace = args[0], [], {}, None
    algo_bytes = {}
    i = 1
    while i < len(args):
        if args[i] == "--flops":
            k, v = args[i + 1].rsplit("=", 1)
            flops[k] = float(v)
            i += 2
        elif args[i] == "--trace":
            trace = args[i + 1]
            i += 2
        elif args[i] == "--bench-json":
            # the JSON line bench.py printed in one of the passes (PCONV_BENCH_TABLE=1): algorithmic
            # flops per launch of every conv kernel, averaged over its launches
            with open(args[i + 1]) as f:
                line = [l for l in f.read().splitlines() if l.startswith("{")][-1]
            per = {}
            for r in json.loads(line).get("roofline_table", []):
                d = per.setdefault(r["kernel"], [0.0, 0])
                d[0] += r["gflop_per_launch"] * 1e9 * r["launches"]
                d[1] += r["launches"]
            for k, (fl, n) in per.items():
                flops[k.replace(", ", ", ")] = fl / n
            # ... and the algorithmic bytes per launch of the HBM-bound kernels (`hbm` rows of the same line), so
            # that bench.py can rescale the counted bytes when its launches are of another size than this pass's
            for r in json.loads(line).get("hbm", []):
                algo_bytes[r["kernel"]] = r["mb_per_launch"] * 1e6
            i += 2
        else:
            dirs.append(args[i])
            i += 1
    merged = defaultdict(dict)
    launches = {}
    for d in dirs:
        for k, counters in read_pass(d).items():
            for c, vals in counters.items():
                merged[k][c] = sum(vals) / len(vals)
                launches[k] = len(vals)
    kernels = {}
    for k, c in merged.items():
        rec = {"launches_per_pass": launches[k], "counters_avg_per_launch": {n: round(v, 3) for n, v in sorted(c.items())}}
        rd = c.get("FETCH_SIZE")
        wr = c.get("WRITE_SIZE")
        if rd is not None:
            rec["hbm_read_bytes_per_launch"] = int(2 * rd * 1024)
        if wr is not None:
            rec["hbm_write_bytes_per_launch"] = int(wr * 1024)
        if rd is not None and wr is not None:
            rec["hbm_bytes_per_launch"] = rec["hbm_read_bytes_per_launch"] + rec["hbm_write_bytes_per_launch"]
        busy, mfma = c.get("SQ_BUSY_CYCLES"), c.get("SQ_VALU_MFMA_BUSY_CYCLES")
        if busy and mfma is not None:
            # SQ_BUSY_CYCLES: summed over the shader engines' SQs (32 on gfx950), cycles any wave
            # is resident; SQ_VALU_MFMA_BUSY_CYCLES: summed over the 256 CUs x 4 SIMDs matrix pipes
            rec["mfma_busy"] = round(mfma / (busy / 32.0 * 1024.0), 4)
        if k in flops:
            rec["algorithmic_flops_per_launch"] = flops[k]
        if k in algo_bytes:
            rec["algorithmic_bytes_per_launch"] = algo_bytes[k]
        kernels[k] = rec
    doc = {"what": "rocprofv3 --pmc passes, averages per launch, per kernel instantiation",
           "passes": [os.path.basename(os.path.normpath(d)) for d in dirs],
           "corrections": "read bytes = 2 x FETCH_SIZE KiB x 1024 (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE exact",
           "mfma_busy_note": "SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 SQs x 1024 SIMDs): share of the resident time the matrix pipes are busy",
           "kernels": kernels}
    with open(out_path, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    print("wrote", out_path, "kernels:", len(kernels))


if __name__ == "__main__":
    main()
