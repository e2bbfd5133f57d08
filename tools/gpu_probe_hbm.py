"""Achieved HBM GB/s of the bandwidth-bound kernels at the 4096x2048 working sizes
(algorithmic bytes = what the op must read + write once)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseudocylindrical_convolution_amd import PCONV  # noqa: E402

W16 = [15., 31., 54., 63., 63., 64., 64., 64., 64., 64., 64., 63., 63., 54., 31., 15.]
DEV = "cuda:0"


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def report(name, seconds, nbytes):
    print("%-44s %8.3f ms  %7.1f GB/s  (%.2f GB algorithmic)" % (name, seconds * 1e3, nbytes / seconds / 1e9, nbytes / 1e9))


def main():
    torch.manual_seed(0)
    ctx = PCONV.PseudoContextOp(16, 20, W16, 0, False)
    # SphereSlice / SphereUslice at full resolution, 3 channels
    frame = torch.rand(1, 3, 2048, 4096, device=DEV)
    sl = PCONV.SphereSliceOp(16, 0, 0, W16, 0, False)
    tiles = sl.forward(frame)[0]
    report("slice 1x3x2048x4096", timed(lambda: sl.forward(frame)), 2 * frame.numel() * 4)
    us = PCONV.SphereUsliceOp(16, 0, 0, W16, 0, False)
    report("uslice 16x3x128x4096", timed(lambda: us.forward(tiles)), 2 * frame.numel() * 4)
    # half-resolution 192-channel activation
    x = torch.rand(16, 192, 64, 2048, device=DEV)
    fill = PCONV.PseudoFillOp(0, 16, 0, 0, ctx.addr(), 0, 0, False)
    valid = float(ctx.widths_host(64, 2048).sum()) / (16 * 2048)
    report("fill (dead columns only) 16x192x64x2048", timed(lambda: fill.forward(x)), (1 - valid) * x.numel() * 4)
    for p in (1, 2):
        pad = PCONV.PseudoPadOp(p, 16, ctx.addr(), 0, False)
        out = pad.forward(x)[0]
        report("pad %d (copying) 16x192x64x2048" % p, timed(lambda: pad.forward(x)), (x.numel() + out.numel()) * 4)
        buf = torch.zeros(16, 192, 64 + 4, 2048 + 4, device=DEV)
        view = buf[:, :, 2:-2, 2:-2]
        view.copy_(x)
        view._pconv_ring = (buf, 2)
        ring_elems = 16 * 192 * (2 * p * (2048 + 2 * p) + 64 * 4 * p)
        report("pad %d (ring only, in place) 16x192x64x2048" % p, timed(lambda: pad.forward_ring(view)), 2 * ring_elems * 4)
    # the codec's batched quarter-scale ring pad (8 frames: 128 tiles of 192 x 32 x 1024, pad 1 in a ring of 1)
    xq = torch.rand(128, 192, 32, 1024, device=DEV)
    pad = PCONV.PseudoPadOp(1, 16, ctx.addr(), 0, False)
    buf = torch.zeros(128, 192, 32 + 2, 1024 + 2, device=DEV)
    view = buf[:, :, 1:-1, 1:-1]
    view.copy_(xq)
    view._pconv_ring = (buf, 1)
    ring_elems = 128 * 192 * (2 * (1024 + 2) + 32 * 2)
    report("pad 1 (ring only, in place) 128x192x32x1024", timed(lambda: pad.forward_ring(view)), 2 * ring_elems * 4)
    y = torch.rand(16, 768, 32, 1024, device=DEV)
    dt = PCONV.DtowOp(2, True, 0, False)
    report("dtow x2 16x768x32x1024", timed(lambda: dt.forward(y)), 2 * y.numel() * 4)


if __name__ == "__main__":
    main()
