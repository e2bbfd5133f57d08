"""Time (and, under rocprofv3 --pmc, count) the dense tile convolution alone.

    python tools/gpu_probe_conv.py [cin cout k stride rows cols reps]
Default: the 192->192 3x3 layer at the half-resolution scale of a 4096x2048 frame.
"""
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseudocylindrical_convolution_amd import PCONV  # noqa: E402


def main():
    a = [int(v) for v in sys.argv[1:]]
    cin, cout, k, stride, rows, cols, reps = (a + [192, 192, 3, 1, 64, 2048, 5][len(a):])[:7]
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    x = torch.randn(16, cin, rows + k - 1, cols + k - 1, device=dev)
    conv = torch.nn.Conv2d(cin, cout, k, stride).to(dev)
    slope = torch.full((cout,), 0.25, device=dev)
    import numpy as np
    from pseudocylindrical_convolution_amd.PCONV_operator import set_weight
    wd = PCONV.tile_widths(np.asarray(set_weight(16, True), dtype=np.float32), 16, rows * 16, cols // stride if stride > 1 else cols)
    limit = torch.from_numpy((wd + (k - 1 if stride == 1 else 0)).astype(np.int32)).to(dev)
    for name, lim in (("all columns", None), ("dead tiles skipped", limit)) * 3:
        for _ in range(2):
            y = PCONV.tile_conv2d(conv, x, conv.weight, conv.bias, stride, slope, lim, 16 if lim is not None else 0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            y = PCONV.tile_conv2d(conv, x, conv.weight, conv.bias, stride, slope, lim, 16 if lim is not None else 0)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        flops = 2.0 * cin * k * k * cout * y.numel() / cout
        print("conv %d->%d k%d s%d out %s, %s: %.3f ms  %.1f TFLOP/s (all-column flops)" % (
            cin, cout, k, stride, tuple(y.shape), name, ms, flops / ms * 1e-9))


if __name__ == "__main__":
    main()
