"""one call of the weight-resident 1x1 kernel on a small shape, with the runtime's error text"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pseudocylindrical_convolution_amd import PCONV
from oracle import pconv_cpu as O
cfgs = [(2, 192, 4, 70, 96), (1, 96, 3, 64, 192)]
os.environ.setdefault("PCONV_CONV1X1", "resident")
if len(sys.argv) > 1:
    cfgs = [tuple(int(v) for v in sys.argv[1:6])]
for tn, cin, h, w, cout in cfgs:
    g = torch.Generator().manual_seed(31)
    x = torch.randn(tn, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) * (1.0 / np.sqrt(cin))
    b = torch.randn(cout, generator=g)
    owner = type("Owner", (), {})()
    sl = torch.rand(cout, generator=g)
    for slope in (None, sl):
        print("launch", (tn, cin, h, w, cout), "slope" if slope is not None else "plain", flush=True)
        y = PCONV.tile_conv2d(owner, x.cuda(), wt.cuda(), b.cuda(), 1, slope.cuda() if slope is not None else None)
        torch.cuda.synchronize()
        print("synced", flush=True)
        ref = O.conv2d_chain(x, wt, b, 1, slope)
        print("equal:", torch.equal(y.cpu(), ref), (y.cpu() - ref).abs().max().item(), flush=True)
