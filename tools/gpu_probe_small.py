"""Time the 192 -> 12 output layer (3x3, depth-to-width store) at a 4096x2048 frame: 16-cout tiles vs the 32-cout tile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pseudocylindrical_convolution_amd import PCONV
from pseudocylindrical_convolution_amd.PCONV_operator import set_weight
dev = torch.device("cuda", 0)
torch.manual_seed(0)
W16 = np.asarray(set_weight(16, True), dtype=np.float32)
x = torch.randn(16, 192, 66, 2050, device=dev)
conv = torch.nn.Conv2d(192, 12, 3).to(dev)
lim = torch.from_numpy(PCONV.tile_widths(W16, 16, 64 * 16, 2048).astype(np.int32)).to(dev)
def run():
    return PCONV.tile_conv2d(conv, x, conv.weight, conv.bias, 1, None, lim, 16, d2w=True)
for _ in range(2): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
px = float(PCONV.tile_widths(W16, 16, 64 * 16, 2048).sum()) * 64
print("3x3 192->12 d2w 64x2048 x16 (PCONV_CONV_SMALL=%s): %.3f ms  %.1f TFLOP/s useful" % (os.environ.get("PCONV_CONV_SMALL", "1"), ms, 2.0 * 192 * 9 * 12 * px / ms * 1e-9))
