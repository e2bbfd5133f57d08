"""Pins for the CPU oracle.  The reference has no vectors for its CUDA kernels
(SURVEY.md 4, 8c), so the restatement in oracle/ is held by the invariants the
algorithm itself implies:
  (i)   entropy encoder -> file -> decoder returns the exact symbols (test_codec_cpu)
  (ii)  every CDF row is strictly increasing from 0 to 65536
  (iii) CDF bin width == 65536 * P(label) as given by the EntropyGmm rate
        (the check sketched in the reference's EntropyGmmTable.py:60-85)
  (iv)  Wtod(Dtow(x)) == x, fill idempotent, slice zero beyond the tile width,
        pad interior == input, uslice(slice(x)) ~ x
  (v)   the wavefront conv equals a dense F.conv2d with conv_mask_v5/v6-masked
        weights on the final causally padded tensor
plus the golden outputs of the reference's importable pure-torch helpers."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import pconv_cpu as O
from pseudocylindrical_convolution_amd.PCONV_operator import set_weight

GOLD = os.path.join(os.path.dirname(__file__), "golden")
W16 = set_weight(16, True)


def rnd(*shape, seed=0):
    return torch.rand(shape, generator=torch.Generator().manual_seed(seed)).contiguous()


def test_dtow_is_a_permutation_and_inverts():
    x = rnd(4, 36, 6, 10, seed=1)
    for s in (2, 3):
        if 36 % (s * s):
            continue
        y = O.DtowOp(s, True).forward(x)[0]
        assert sorted(y.flatten().tolist()) == sorted(x.flatten().tolist())
        assert torch.equal(O.DtowOp(s, False).forward(y.clone())[0], x)
    # matches torch pixel_shuffle (same channel -> sub-position convention)
    assert torch.equal(O.DtowOp(2, True).forward(x)[0], F.pixel_shuffle(x, 2))


def test_slice_zero_tail_fill_idempotent_pad_interior():
    x = rnd(1, 2, 256, 512, seed=2)
    t = O.SphereSliceOp(16, 0, 0, W16).forward(x)[0]
    wd = O.widths_v3(np.array(W16, np.float32), 16, 256, 512)
    for i in range(16):
        assert t[i, :, :, int(wd[i]):].abs().sum() == 0
        assert t[i, :, :, :int(wd[i])].abs().min() > 0
    ctx = O.PseudoContextOp(16, 20, W16)
    f = O.PseudoFillOp(0, 16, 0, 0, ctx.addr(), 0)
    once = f.forward(rnd(16, 2, 16, 512, seed=3))[0].clone()
    assert torch.equal(f.forward(once.clone())[0], once)
    p = O.PseudoPadOp(2, 16, ctx.addr()).forward(once.clone())[0]
    for i in range(16):
        v = int(wd[i])
        assert torch.equal(p[i, :, 2:-2, 2:2 + v], once[i, :, :, :v])
        # circular wrap of the valid width
        assert torch.equal(p[i, :, :, 0:2], p[i, :, :, v:v + 2])
        assert torch.equal(p[i, :, :, v + 2:v + 4], p[i, :, :, 2:4])
        assert p[i, :, :, v + 4:].abs().sum() == 0
    # equatorial tiles (full width on both sides): halo rows are the neighbour's rows
    assert torch.allclose(p[6, :, 0:2, 2:-2], once[5, :, -2:, :], atol=1e-6)
    assert torch.allclose(p[6, :, -2:, 2:-2], once[7, :, :2, :], atol=1e-6)


def test_slice_then_uslice_reconstructs_smooth_images():
    yy, xx = torch.meshgrid(torch.linspace(0, 1, 256), torch.linspace(0, 1, 512), indexing="ij")
    img = (0.5 + 0.3 * torch.sin(6.2832 * xx) * torch.cos(3.1416 * yy)).view(1, 1, 256, 512).contiguous()
    t = O.SphereSliceOp(16, 0, 0, W16).forward(img)[0]
    back = O.SphereUsliceOp(16, 0, 0, W16).forward(t.clone())[0]
    assert (back - img).abs().max() < 2e-3


def test_cdf_rows_and_rate_identity():
    n = 3000
    g = torch.Generator().manual_seed(5)
    raw = torch.randn(3, 3, 30, 100, generator=g)
    raw[1] = raw[1].abs() * 2 + 0.05
    raw[2] = raw[2] * 3
    tnum = torch.tensor([n], dtype=torch.int32)
    for det in (False, True):
        O.set_detmath(det)
        data = raw.clone()
        tab = O.EntropyGmmTableOp(8, 3.5, 3, 65536, 1e-6).forward_batch(data, tnum)[0][:n]
        assert (tab[:, 0] == 0).all() and (tab[:, 8] == 65536).all()
        assert (tab[:, 1:] - tab[:, :-1] >= 1).all()
        # rate identity: width of bin k == 65536 * exp(-EntropyGmm(label = k - 3.5)), interior bins
        flat = data.view(3, -1)
        wt, dl, mu = (flat[i][:n * 3].view(n, 3).contiguous() for i in range(3))
        for k in (2, 3, 4, 5):
            label = torch.full((n, 1), k - 3.5)
            loss = O.EntropyGmmOp(3, 0).forward(wt, dl, mu, label)[0]
            width = (tab[:, k + 1] - tab[:, k]).double()
            pred = 65536.0 * torch.exp(-loss.double())
            assert (width - pred).abs().max() < 4.0   # rounding of two table entries + repair
    O.set_detmath(True)


def test_detmath_close_to_libm():
    x = torch.linspace(-6, 6, 200001)
    xs = x.numpy().astype(np.float32)
    import ctypes
    lib = O.lib()
    # probe through the gmm loss: erf differences show up as loss differences; direct check via tables
    O.set_detmath(True)
    raw = torch.zeros(3, 3, 1, 1)
    raw[1] = 1.0
    tnum = torch.tensor([1], dtype=torch.int32)
    a = O.EntropyGmmTableOp(8, 3.5, 3, 65536, 1e-6).forward_batch(raw.clone(), tnum)[0][:1]
    O.set_detmath(False)
    b = O.EntropyGmmTableOp(8, 3.5, 3, 65536, 1e-6).forward_batch(raw.clone(), tnum)[0][:1]
    O.set_detmath(True)
    assert (a - b).abs().max() <= 1


@pytest.mark.parametrize("constrain,cin_g", [(5, 1), (6, 3)])
def test_wavefront_conv_equals_dense_masked_conv(constrain, cin_g):
    """(v): run one masked layer over the whole wavefront, then compare every valid
    position with F.conv2d(masked weight) on the final padded input."""
    G, h, w, nimg = 4, 2, 64, 1
    ctx = O.EntropyContextOp(16, 18, W16)
    ctx.start_context(w)
    g = torch.Generator().manual_seed(11)
    cin, cout = G * cin_g, G * 3
    x_full = torch.randn(nimg * 16, cin, h, w, generator=g)
    fill = O.PseudoFillOp(0, 16, 0, 0, ctx.addr(), 2)
    x_full = fill.forward(x_full)[0]
    weight = torch.randn(1, cout, cin, 5, 5, generator=g) * 0.1
    bias = torch.randn(1, cout, generator=g) * 0.1
    pad = O.EntropyCtxPadRun2Op(2, 16, G, False, ctx.addr())
    conv = O.EntropyConv2Op(16, cin, G, cout, 5, constrain, 2, 0, ctx.addr())
    # the padded input is complete from the start; halos are filled step by step
    xp = torch.zeros(nimg * 16, cin, h + 4, w + 4)
    xp[:, :, 2:-2, 2:-2] = x_full
    for _ in range(16 * h + w + G - 2):
        y = conv.forward_batch(pad.forward(xp)[0], weight, bias)[0]
    masked = weight[0].clone()
    O.MaskConstrainOp(constrain, G).forward(masked)
    dense = F.conv2d(xp, masked, bias[0])
    wd = O.widths_v3(np.array(W16, np.float32), 16, 16 * h, w)
    for t in range(16):
        v = int(wd[t])
        assert (y[t::16, :, :, :v] - dense[t::16, :, :, :v]).abs().max() < 1e-4
    # and the summation orders of the oracle -- 0: the reference's 128-thread tree, 1: the product's (the
    # default: tap-major, 64 lanes), 2: the causal-compact order of the round-4 band-kernel experiment
    # (only the unmasked entries are enumerated) -- agree to float accuracy
    assert O.CONV_ORDER == 1
    for order in (0, 2):
        O.CONV_ORDER = order
        try:
            conv0 = O.EntropyConv2Op(16, cin, G, cout, 5, constrain, 2, 0, ctx.addr())
            pad0 = O.EntropyCtxPadRun2Op(2, 16, G, False, ctx.addr())
            for _ in range(16 * h + w + G - 2):
                y0 = conv0.forward_batch(pad0.forward(xp)[0], weight, bias)[0]
        finally:
            O.CONV_ORDER = 1
        assert (y0 - y).abs().max() < 1e-5


def test_quantiser_levels():
    x = rnd(16, 4, 2, 64, seed=7)
    weight = torch.zeros(4, 8)
    weight[:, 0] = 1. / 9
    weight[:, 1:] = float(np.log(1. / 9))
    ctx = O.PseudoContextOp(16, 20, W16)
    val, idx = O.PseudoQuantOp(4, 8, 16, 0.9, 100, 2, 0.1, ctx.addr()).forward(x, weight, torch.zeros(4, 8), False)
    wd = O.widths_v3(np.array(W16, np.float32), 16, 32, 64)
    levels = 1. / 9 + np.arange(8) / 9.0
    for t in range(16):
        v = int(wd[t])
        assert idx[t, :, :, v:].abs().sum() == 0 and val[t, :, :, v:].abs().sum() == 0
        nearest = np.abs(x[t, :, :, :v].numpy()[..., None] - levels).argmin(-1)
        assert (idx[t, :, :, :v].numpy() == nearest).all()
        assert np.abs(val[t, :, :, :v].numpy() - levels[nearest]).max() < 1e-6
    deq = O.PseudoDQuantOp(16, 4, 8, ctx.addr()).forward(idx.clone(), weight)[0]
    assert torch.allclose(deq, val, atol=1e-6)


def test_conv_chain_oracle_close_to_torch():
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 24, 7, 33, generator=g)
    wt = torch.randn(16, 24, 3, 3, generator=g) * 0.1
    b = torch.randn(16, generator=g)
    for stride in (1, 2):
        assert (O.conv2d_chain(x, wt, b, stride) - F.conv2d(x, wt, b, stride)).abs().max() < 1e-5


def test_pure_torch_helpers_match_reference_golden():
    from pseudocylindrical_convolution_amd.PCONV_operator import LowerBound, SSIM, Extract
    d = np.load(os.path.join(GOLD, "torch_helpers.npz"))
    out = LowerBound.apply(torch.from_numpy(d["lb_x"]), torch.from_numpy(d["lb_bound"]))
    assert torch.equal(out, torch.from_numpy(d["lb_out"]))
    s = SSIM(11, 3)(torch.from_numpy(d["ssim_a"]), torch.from_numpy(d["ssim_b"]))
    assert abs(s.item() - float(d["ssim_out"])) < 1e-6
    assert torch.equal(Extract(5)(torch.from_numpy(d["ex_in"])), torch.from_numpy(d["ex_out"]))
