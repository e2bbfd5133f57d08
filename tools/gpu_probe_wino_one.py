"""One Winograd launch shape, a few repetitions (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pseudocylindrical_convolution_amd import PCONV
from pseudocylindrical_convolution_amd.PCONV_operator import set_weight
dev = torch.device("cuda", 0)
torch.manual_seed(0)
W16 = np.asarray(set_weight(16, True), dtype=np.float32)
tn, cin, cout, rows, cols = 16, 192, 192, 64, 2048
x = torch.randn(tn, cin, rows + 2, cols + 2, device=dev)
conv = torch.nn.Conv2d(cin, cout, 3).to(dev)
sl = torch.rand(cout, device=dev)
r = torch.randn(tn, cout, rows, cols, device=dev)
lim = torch.from_numpy(PCONV.tile_widths(W16, 16, rows * 16, cols).astype(np.int32)).to(dev)
os.environ["PCONV_CONV3X3"] = sys.argv[1] if len(sys.argv) > 1 else "wino"
for _ in range(4):
    PCONV.tile_conv2d(conv, x, conv.weight, conv.bias, 1, sl, lim, 16, residual=r, trim=True, ring=2)
torch.cuda.synchronize()
