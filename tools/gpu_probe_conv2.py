"""Where does the tile convolution's time go?  (DVFS / dispatch diagnostics)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseudocylindrical_convolution_amd import PCONV  # noqa: E402
from pseudocylindrical_convolution_amd.PCONV_operator import set_weight  # noqa: E402


def run(name, x, conv, slope, lim, npart, reps=10):
    for _ in range(2):
        y = PCONV.tile_conv2d(conv, x, conv.weight, conv.bias, 1, slope, lim, npart)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        y = PCONV.tile_conv2d(conv, x, conv.weight, conv.bias, 1, slope, lim, npart)
    e1.record()
    torch.cuda.synchronize()
    print("%-46s %.3f ms" % (name, e0.elapsed_time(e1) / reps), flush=True)


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    cin = cout = 192
    rows, cols, k = 64, 2048, 3
    x = torch.randn(16, cin, rows + 2, cols + 2, device=dev)
    conv = torch.nn.Conv2d(cin, cout, k).to(dev)
    slope = torch.full((cout,), 0.25, device=dev)
    wd = PCONV.tile_widths(np.asarray(set_weight(16, True), dtype=np.float32), 16, rows * 16, cols)
    limit = torch.from_numpy((wd + 2).astype(np.int32)).to(dev)
    run("16 tiles, all columns", x, conv, slope, None, 0)
    run("16 tiles, dead skipped (83% live)", x, conv, slope, limit, 16)
    run("13 tiles, all columns (81% of the grid)", x[:13].contiguous(), conv, slope, None, 0)
    run("8 tiles, all columns", x[:8].contiguous(), conv, slope, None, 0)
    half = torch.full((16,), 1026, dtype=torch.int32, device=dev)
    run("16 tiles, limit 1026 everywhere (50% live)", x, conv, slope, half, 16)
    x0 = torch.zeros_like(x)
    run("16 tiles, all columns, zero input", x0, conv, slope, None, 0)
    run("16 tiles, dead skipped, zero input", x0, conv, slope, limit, 16)
    run("16 tiles, all columns (again)", x, conv, slope, None, 0)


if __name__ == "__main__":
    main()
