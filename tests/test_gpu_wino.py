"""GPU: the Winograd F(2x2, 3x3) tile convolution (csrc/wino.hip), the product's default for the 3x3
stride-1 layers, against

  * the direct fp32-MFMA kernel (the k-ascending fmaf chain the oracle restates bit for bit), and
  * a float64 convolution on the CPU (so that the bound is on the error of each, not on their
    difference only);

every epilogue it takes (bias, PReLU, residual + trim, the Dtow store), dense tensors and views into
padded buffers, ragged edges, dead-column blocks, batched tile stacks.  Bound: 2e-5 absolute on
outputs of unit scale -- the north-star tolerance is 1e-4 over the whole transform."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def P():
    from pseudocylindrical_convolution_amd import PCONV
    return PCONV


def conv(monkeypatch, mode, *args, **kwargs):
    monkeypatch.setenv("PCONV_CONV3X3", mode)
    owner = type("Owner", (), {})()
    return P().tile_conv2d(owner, *args, **kwargs)


def data(tn, cin, h, w, cout, seed=5):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(tn, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (1.0 / np.sqrt(cin * 9))
    b = torch.randn(cout, generator=g)
    sl = torch.rand(cout, generator=g)
    return x, wt, b, sl


@pytest.mark.parametrize("cfg", [
    # tn, cin, h, w, cout
    (2, 192, 6, 70, 192), (1, 96, 6, 66, 96), (16, 192, 4, 130, 192), (1, 192, 10, 258, 768), (3, 32, 8, 40, 40), (2, 16, 6, 36, 32),
    (2, 192, 68, 260, 192), (1, 96, 12, 1028, 96),
])
def test_wino_matches_direct_kernel_and_float64(cfg, hip_backend, monkeypatch):
    tn, cin, h, w, cout = cfg
    x, wt, b, sl = data(*cfg)
    assert P()._native.hip_lib().pconv_wino_supported(cin, h, w, cout, 0) == 1
    ref64 = torch.nn.functional.conv2d(x.double(), wt.double(), b.double())
    for slope in (None, sl):
        want = ref64 if slope is None else torch.where(ref64 < 0, ref64 * slope.double().view(1, -1, 1, 1), ref64)
        yd = conv(monkeypatch, "direct", x.to(DEV), wt.to(DEV), b.to(DEV), 1, slope.to(DEV) if slope is not None else None).cpu()
        yw = conv(monkeypatch, "wino", x.to(DEV), wt.to(DEV), b.to(DEV), 1, slope.to(DEV) if slope is not None else None).cpu()
        assert yw.shape == yd.shape
        ed, ew = (yd.double() - want).abs().max().item(), (yw.double() - want).abs().max().item()
        assert ed < 2e-5 and ew < 2e-5, "direct %g, winograd %g from float64" % (ed, ew)
        assert (yw - yd).abs().max().item() < 2e-5


def test_wino_is_the_default_and_direct_stays_selectable(hip_backend, monkeypatch):
    """default (round 4): F(4x2, 3x3) for the layers with 64-multiple couts it takes, F(2x2, 3x3) for the rest of the
    3x3 stride-1 layers (a 96-cout layer here)"""
    x, wt, b, _ = data(1, 96, 6, 66, 96)
    monkeypatch.delenv("PCONV_CONV3X3", raising=False)
    assert P().conv3x3_mode() == "wino42"
    owner = type("Owner", (), {})()
    rec = type("Probe", (), {"records": []})()
    monkeypatch.setattr(P(), "conv_probe", rec)
    P().tile_conv2d(owner, x.to(DEV), wt.to(DEV), b.to(DEV), 1)
    monkeypatch.setenv("PCONV_CONV3X3", "direct")
    P().tile_conv2d(owner, x.to(DEV), wt.to(DEV), b.to(DEV), 1)
    assert [r[0].split("<")[0] for r in rec.records] == ["wino_conv3x3_kernel", "conv_mfma_kernel"]
    # what it does not take goes to the direct kernels: stride 2, odd output size (conv_mfma_kernel), the 12-cout
    # output layer (16-cout tiles: conv_small_kernel)
    monkeypatch.setenv("PCONV_CONV3X3", "wino")
    rec.records.clear()
    P().tile_conv2d(owner, x.to(DEV), wt.to(DEV), b.to(DEV), 2)
    P().tile_conv2d(owner, x[:, :, :5].contiguous().to(DEV), wt.to(DEV), b.to(DEV), 1)
    P().tile_conv2d(owner, x.to(DEV), wt[:12].contiguous().to(DEV), b[:12].contiguous().to(DEV), 1)
    assert [r[0].split("<")[0] for r in rec.records] == ["conv_mfma_kernel", "conv_mfma_kernel", "conv_small_kernel"]


@pytest.mark.parametrize("cfg", [(16, 192, 6, 70, 192), (32, 96, 6, 134, 96), (16, 192, 10, 262, 192)])
def test_wino_epilogue_views_and_dead_columns(cfg, hip_backend, monkeypatch):
    """residual + trim + PReLU, input / residual / output inside padded buffers, per-tile column limits
    (dead 64-column blocks are zeros, trimmed columns are zeros, everything else == the direct kernel's)"""
    tn, cin, h, w, cout = cfg
    x, wt, b, sl = data(*cfg, seed=8)
    ho, wo = h - 2, w - 2
    g = torch.Generator().manual_seed(9)
    res = torch.randn(tn, cout, ho, wo, generator=g)
    limit = torch.tensor([wo, 40, 64, 3, 65, 128, wo - 1, 1] * 2, dtype=torch.int32).to(DEV)

    def inside(t, p):
        buf = torch.full((t.shape[0], t.shape[1], t.shape[2] + 2 * p, t.shape[3] + 2 * p), 7.0, device=DEV)
        buf[:, :, p:-p, p:-p] = t.to(DEV)
        return buf[:, :, p:-p, p:-p]

    args = (wt.to(DEV), b.to(DEV), 1, sl.to(DEV), limit, 16)
    for (xin, rin, ring) in ((x.to(DEV), res.to(DEV), 0), (inside(x, 2), inside(res, 2), 2), (inside(x, 1), res.to(DEV), 2)):
        yd = conv(monkeypatch, "direct", xin, *args, residual=rin, trim=True, ring=ring)
        yw = conv(monkeypatch, "wino", xin, *args, residual=rin, trim=True, ring=ring)
        assert (yw - yd).abs().max().item() < 2e-5
        for t in range(tn):
            lim = int(limit[t % 16])
            assert yw[t, :, :, lim:].abs().max().item() == 0 if lim < wo else True
        if ring:
            buf = yw._pconv_ring[0]
            assert tuple(buf.shape) == (tn, cout, ho + 2 * ring, wo + 2 * ring)
    # no trim: a live block is computed to its end, a dead block is zeros
    yd = conv(monkeypatch, "direct", x.to(DEV), wt.to(DEV), b.to(DEV), 1, None, limit, 16)
    yw = conv(monkeypatch, "wino", x.to(DEV), wt.to(DEV), b.to(DEV), 1, None, limit, 16)
    assert (yw - yd).abs().max().item() < 2e-5


@pytest.mark.parametrize("cfg", [(16, 192, 6, 70, 768), (2, 96, 10, 134, 192)])
def test_wino_depth_to_width_store(cfg, hip_backend, monkeypatch):
    tn, cin, h, w, cout = cfg
    x, wt, b, sl = data(*cfg, seed=11)
    limit = torch.tensor([w - 2, 40, 64, 3] * 4, dtype=torch.int32).to(DEV)
    for slope in (None, sl.to(DEV)):
        plain = conv(monkeypatch, "wino", x.to(DEV), wt.to(DEV), b.to(DEV), 1, slope, limit, 16)
        shuffled = P().DtowOp(2, True, 0, False).forward(plain)[0].clone()
        for ring in (0, 2):
            fused = conv(monkeypatch, "wino", x.to(DEV), wt.to(DEV), b.to(DEV), 1, slope, limit, 16, d2w=True, ring=ring)
            assert tuple(fused.shape) == tuple(shuffled.shape)
            assert torch.equal(fused, shuffled)                     # same arithmetic, another store
        direct = conv(monkeypatch, "direct", x.to(DEV), wt.to(DEV), b.to(DEV), 1, slope, limit, 16, d2w=True)
        assert (direct - shuffled).abs().max().item() < 2e-5


def test_wino_follows_reloaded_weights(hip_backend, monkeypatch):
    """the packed U = G g Gt is cached per parameter version"""
    monkeypatch.setenv("PCONV_CONV3X3", "wino")
    m = torch.nn.Conv2d(96, 96, 3).to(DEV)
    x = torch.randn(1, 96, 6, 66, device=DEV)
    y0 = P().tile_conv2d(m, x, m.weight, m.bias, 1).clone()
    with torch.no_grad():
        m.weight.mul_(2.0)
    y1 = P().tile_conv2d(m, x, m.weight, m.bias, 1).clone()
    ref = torch.nn.functional.conv2d(x, m.weight, m.bias)
    assert (y1 - ref).abs().max().item() < 1e-4 and (y1 - y0).abs().max().item() > 1e-3


@pytest.mark.parametrize("xscale,wscale", [(1e3, 1.0), (1e-3, 1.0), (1.0, 1e3), (1e3, 1e-3), (30.0, 30.0)])
def test_wino_relative_error_at_other_scales(xscale, wscale, hip_backend, monkeypatch):
    """A trained model's activations are not unit scale (GDN outputs, attention products): the bound is
    RELATIVE here.  Inputs x xscale, weights of mixed scale (every fourth output channel x 100, every
    third input channel x 0.01, on top of wscale), bias 0: against a float64 convolution the error of
    every output stays under 1e-5 (Winograd; measured 2.9e-6) / 2e-5 (the direct kernel's 1728-term fmaf
    chain; measured 7.9e-6) of the layer's output scale sqrt(sum_k w^2 x^2), at every scale alike -- fp32
    arithmetic is scale-free until it overflows -- and Winograd is never worse than 3 x the direct kernel."""
    tn, cin, h, w, cout = 2, 192, 10, 134, 192
    g = torch.Generator().manual_seed(21)
    x = torch.randn(tn, cin, h, w, generator=g) * xscale
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (wscale / np.sqrt(cin * 9))
    wt[::4] *= 100.0
    wt[:, ::3] *= 0.01
    b = torch.zeros(cout)
    ref64 = torch.nn.functional.conv2d(x.double(), wt.double())
    # output scale per (tile, cout, pixel): the l2 norm of the products that are summed
    scale = torch.nn.functional.conv2d(x.double() ** 2, wt.double() ** 2).sqrt()
    yd = conv(monkeypatch, "direct", x.to(DEV), wt.to(DEV), b.to(DEV), 1).cpu().double()
    yw = conv(monkeypatch, "wino", x.to(DEV), wt.to(DEV), b.to(DEV), 1).cpu().double()
    rd = ((yd - ref64).abs() / scale).max().item()
    rw = ((yw - ref64).abs() / scale).max().item()
    assert rd < 2e-5 and rw < 1e-5, "relative error: direct %g, winograd %g" % (rd, rw)
    assert rw < 3 * rd + 1e-7, "winograd %g vs direct %g" % (rw, rd)
    assert torch.isfinite(yw).all()


def test_wino_fallbacks_are_counted(hip_backend, monkeypatch):
    """a 3x3 stride-1 layer Winograd was selected for but does not take (cin % 16 != 0, 12 couts) goes to
    the direct kernel and is counted in PCONV.conv_fallbacks; a layer it takes is not"""
    monkeypatch.setenv("PCONV_CONV3X3", "wino")
    P().conv_fallbacks.clear()
    owner = type("Owner", (), {})()
    x, wt, b, _ = data(1, 24, 6, 66, 40)
    P().tile_conv2d(owner, x.to(DEV), wt.to(DEV), b.to(DEV), 1)
    assert P().conv_fallbacks == {(24, 6, 66, 40, False): 1}
    x, wt, b, _ = data(1, 96, 6, 66, 96)
    P().tile_conv2d(type("Owner", (), {})(), x.to(DEV), wt.to(DEV), b.to(DEV), 1)
    assert len(P().conv_fallbacks) == 1
    assert P()._native.hip_lib().pconv_wino_supported(8, 6, 66, 96, 0) == 0      # the header's contract: cin % 16
    assert P()._native.hip_lib().pconv_wino_supported(16, 6, 66, 32, 0) == 1


@pytest.mark.parametrize("cfg", [(16, 192, 4, 262, 192), (3, 96, 4, 70, 96), (32, 192, 4, 1030, 192), (2, 48, 4, 130, 64)])
def test_two_row_launches_take_the_flat_tile_and_write_the_same_bits(cfg, hip_backend, monkeypatch):
    """(r6) csrc/wino_flat.hip = wino.hip compiled with a 2-row x 128-column workgroup tile, for launches of exactly two
    output rows (the remainders of the row split): the same operations per output in the same order, so the SAME BITS as
    the 4 x 64 tile -- plain, residual + trim + PReLU with column limits, inside padded buffers, and the Dtow store"""
    tn, cin, h, w, cout = cfg
    x, wt, b, sl = data(*cfg, seed=31)
    ho, wo = h - 2, w - 2
    assert ho == 2
    res = torch.randn(tn, cout, ho, wo, generator=torch.Generator().manual_seed(32))
    limit = torch.tensor([wo, 40, 64, 3, 65, 128, wo - 1, 1] * 2, dtype=torch.int32).to(DEV)

    def inside(t, p):
        buf = torch.full((t.shape[0], t.shape[1], t.shape[2] + 2 * p, t.shape[3] + 2 * p), 7.0, device=DEV)
        buf[:, :, p:-p, p:-p] = t.to(DEV)
        return buf[:, :, p:-p, p:-p]

    def run(flat, xin, **kw):
        monkeypatch.setattr(P(), "WINO_FLAT_REMAINDER", flat)
        return conv(monkeypatch, "wino", xin, wt.to(DEV), b.to(DEV), 1, **kw)

    ref64 = torch.nn.functional.conv2d(x.double(), wt.double(), b.double())
    a, f = run(False, x.to(DEV)), run(True, x.to(DEV))
    assert torch.equal(a, f) and (f.cpu().double() - ref64).abs().max().item() < 2e-5
    for (xin, rin, ring) in ((x.to(DEV), res.to(DEV), 0), (inside(x, 2), inside(res, 2), 2)):
        kw = dict(slope=sl.to(DEV), col_limit=limit, npart=16, residual=rin, trim=True, ring=ring)
        a, f = run(False, xin, **kw), run(True, xin, **kw)
        assert torch.equal(a, f)
        d = conv(monkeypatch, "direct", xin, wt.to(DEV), b.to(DEV), 1, **kw)
        assert (f - d).abs().max().item() < 2e-5
    if cout % 4 == 0:
        a, f = run(False, x.to(DEV), slope=sl.to(DEV), d2w=True), run(True, x.to(DEV), slope=sl.to(DEV), d2w=True)
        assert a.shape == (tn, cout // 4, 2 * ho, 2 * wo) and torch.equal(a, f)
