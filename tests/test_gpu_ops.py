"""GPU parity of every hot-path op: HIP kernel (through the C ABI) vs the CPU
oracle on the same seeded inputs.  Integer / index work and every gather/lerp
kernel must match bit for bit (both sides evaluate the same IEEE operations in
the same order, contraction off); the dense convolution matches the fmaf-chain
oracle bit for bit and torch's CPU conv within 1e-4 (north_star tolerance)."""
import numpy as np
import pytest
import torch

from oracle import pconv_cpu as O

pytestmark = pytest.mark.gpu

W16 = [15., 31., 54., 63., 63., 64., 64., 64., 64., 64., 64., 63., 63., 54., 31., 15.]
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _detmath():
    O.set_detmath(True)
    yield


@pytest.fixture(autouse=True)
def _direct_conv(monkeypatch):
    """this file pins the DIRECT tile convolution (one k-ascending fmaf chain per output, bit-exact
    against the oracle's restatement); the Winograd form of the 3x3 stride-1 layers, the product's
    default, has its own tests in tests/test_gpu_wino.py"""
    monkeypatch.setenv("PCONV_CONV3X3", "direct")


def P():
    from pseudocylindrical_convolution_amd import PCONV
    return PCONV


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * scale).contiguous()


def same(a_gpu, b_cpu):
    a = a_gpu.detach().cpu()
    assert a.shape == b_cpu.shape
    assert torch.equal(a, b_cpu), "max abs diff %g" % (a - b_cpu).abs().max().item()


@pytest.mark.parametrize("shape,pad", [((1, 3, 512, 1024), 0), ((2, 2, 256, 512), 0), ((1, 1, 256, 320), 1)])
def test_sphere_slice(shape, pad):
    x = rnd(*shape, seed=1)
    g = P().SphereSliceOp(16, 0, pad, W16, 0, False).forward(x.to(DEV))[0]
    c = O.SphereSliceOp(16, 0, pad, W16).forward(x)[0]
    if pad:
        p = pad
        same(g[:, :, p:-p, p:-p], c[:, :, p:-p, p:-p].contiguous())
    else:
        same(g, c)
    # columns at or beyond the tile width are exactly zero
    widths = O.widths_v3(W16, 16, shape[2], shape[3])
    gi = g[:, :, pad:g.shape[2] - pad, pad:g.shape[3] - pad].cpu()
    for t in range(16):
        assert gi[t::16, :, :, int(widths[t]):].abs().max().item() == 0 if widths[t] < shape[3] else True


@pytest.mark.parametrize("shape,pad", [((16, 3, 32, 1024), 0), ((32, 2, 16, 512), 0), ((16, 2, 18, 516), 2)])
def test_sphere_uslice(shape, pad):
    x = rnd(*shape, seed=2)
    g = P().SphereUsliceOp(16, 0, pad, W16, 0, False).forward(x.to(DEV))[0]
    c = O.SphereUsliceOp(16, 0, pad, W16).forward(x)[0]
    same(g, c)


@pytest.mark.parametrize("shape,pad", [((16, 4, 16, 512), 1), ((16, 3, 8, 256), 2), ((32, 2, 2, 64), 2),
                                       ((16, 2, 32, 1024), 1)])
def test_pseudo_pad(shape, pad):
    x = rnd(*shape, seed=3)
    gctx = P().PseudoContextOp(16, 20, W16, 0, False)
    octx = O.PseudoContextOp(16, 20, W16)
    # zero the dead columns first, as the network does
    xg = P().PseudoFillOp(0, 16, 0, 0, gctx.addr(), 0, 0, False).forward(x.clone().to(DEV))[0]
    xc = O.PseudoFillOp(0, 16, 0, 0, octx.addr(), 0).forward(x.clone())[0]
    same(xg, xc)
    g = P().PseudoPadOp(pad, 16, gctx.addr(), 0, False).forward(xg)[0]
    c = O.PseudoPadOp(pad, 16, octx.addr()).forward(xc)[0]
    same(g, c)
    # inside each tile's valid width the interior of the padded tensor is the input
    widths = O.widths_v3(W16, 16, 16 * shape[2], shape[3])
    inner = g[:, :, pad:-pad, pad:-pad].cpu()
    for t in range(16):
        v = int(widths[t])
        assert torch.equal(inner[t::16, :, :, :v], xc[t::16, :, :, :v])


def test_pseudo_pad_unmasked_input():
    # also exact when the dead columns hold garbage (pad must not read them)
    x = rnd(16, 2, 4, 128, seed=4)
    gctx = P().PseudoContextOp(16, 20, W16, 0, False)
    octx = O.PseudoContextOp(16, 20, W16)
    same(P().PseudoPadOp(2, 16, gctx.addr(), 0, False).forward(x.to(DEV))[0],
         O.PseudoPadOp(2, 16, octx.addr()).forward(x)[0])


@pytest.mark.parametrize("pad,trim,fvalue", [(0, 0, 0), (2, 0, 0), (2, 1, 3), (1, 1, -1)])
def test_pseudo_fill(pad, trim, fvalue):
    x = rnd(32, 3, 10, 132, seed=5)
    gctx = P().PseudoContextOp(16, 20, W16, 0, False)
    octx = O.PseudoContextOp(16, 20, W16)
    op = P().PseudoFillOp(pad, 16, fvalue, trim, gctx.addr(), 0, 0, False)
    xg = x.clone().to(DEV)
    g = op.forward(xg)[0]
    assert g.data_ptr() == xg.data_ptr()  # in place, returns its input
    c = O.PseudoFillOp(pad, 16, fvalue, trim, octx.addr(), 0).forward(x.clone())[0]
    same(g, c)
    same(op.forward(g)[0], c)  # idempotent


@pytest.mark.parametrize("shape,stride", [((16, 56, 2, 64), 2), ((3, 36, 5, 7), 3), ((16, 768, 4, 32), 2),
                                          ((2, 12, 6, 10), 2)])
def test_dtow_roundtrip(shape, stride):
    x = rnd(*shape, seed=6)
    d = P().DtowOp(stride, True, 0, False)
    w = P().DtowOp(stride, False, 0, False)
    g = d.forward(x.to(DEV))[0]
    same(g, O.DtowOp(stride, True).forward(x)[0])
    back = w.forward(g)[0]
    same(back, x)
    same(back, O.DtowOp(stride, False).forward(g.cpu())[0])


def test_quant_dquant():
    x = rnd(16, 192, 2, 64, seed=7)
    weight = torch.zeros(192, 8)
    weight[:, 0] = 1. / 9
    weight[:, 1:] = float(np.log(1. / 9))
    weight += rnd(192, 8, seed=8) * 0.05
    count = torch.zeros(192, 8)
    gctx = P().PseudoContextOp(16, 20, W16, 0, False)
    octx = O.PseudoContextOp(16, 20, W16)
    gv, gi = P().PseudoQuantOp(192, 8, 16, 0.9, 100, 2, 0.1, gctx.addr(), 0, False).forward(
        x.to(DEV), weight.to(DEV), count.to(DEV), False)
    cv, ci = O.PseudoQuantOp(192, 8, 16, 0.9, 100, 2, 0.1, octx.addr()).forward(x, weight, count, False)
    same(gi, ci)
    same(gv, cv)
    assert ci.max() <= 7 and ci.min() >= 0
    sub = ci[:, :56].contiguous()
    gd = P().PseudoDQuantOp(16, 192, 8, gctx.addr(), 0, False).forward(sub.to(DEV), weight.to(DEV))[0]
    cd = O.PseudoDQuantOp(16, 192, 8, octx.addr()).forward(sub, weight)[0]
    same(gd, cd)


def test_quant_dquant_sixteen_byte_forms():
    """(r6) eval mode under no_grad -- the codec's call -- takes the kernels' 16-byte forms (8 levels, 4 columns per
    thread, the channel's levels in registers): same values and indices as the oracle, ragged tile widths (the valid
    width of a tile ends inside a quad), values on and beyond the outermost levels; the per-call histogram stays zero"""
    x = rnd(32, 192, 4, 128, seed=21) * 1.5
    x[0, 0, 0, :8] = torch.tensor([-5.0, 5.0, 0.0, 1. / 9, 2. / 9, 0.5, 0.999, 1.0])
    weight = torch.zeros(192, 8)
    weight[:, 0] = 1. / 9
    weight[:, 1:] = float(np.log(1. / 9))
    weight += rnd(192, 8, seed=22) * 0.2
    count = torch.zeros(192, 8)
    gctx = P().PseudoContextOp(16, 20, W16, 0, False)
    octx = O.PseudoContextOp(16, 20, W16)
    gop = P().PseudoQuantOp(192, 8, 16, 0.9, 100, 2, 0.1, gctx.addr(), 0, False)
    with torch.no_grad():
        gv, gi = gop.forward(x.to(DEV), weight.to(DEV), count.to(DEV), False)
        assert float(gop.count_data_.abs().sum()) == 0.0
        cv, ci = O.PseudoQuantOp(192, 8, 16, 0.9, 100, 2, 0.1, octx.addr()).forward(x, weight, count, False)
        same(gi, ci)
        same(gv, cv)
        assert set(ci.unique().tolist()) == set(float(k) for k in range(8))
        sub = ci[:, :56].contiguous()
        gd = P().PseudoDQuantOp(16, 192, 8, gctx.addr(), 0, False).forward(sub.to(DEV), weight.to(DEV))[0]
        cd = O.PseudoDQuantOp(16, 192, 8, octx.addr()).forward(sub, weight)[0]
        same(gd, cd)
    # ... and with gradients enabled the scalar form with the histogram, same values
    gv2, gi2 = gop.forward(x.to(DEV), weight.to(DEV), count.to(DEV), False)
    same(gi2, ci)
    same(gv2, cv)
    assert float(gop.count_data_.sum()) < 0.0


def test_projects():
    x = rnd(1, 3, 512, 1024, seed=9)
    th = [-0.5, 0, 0.5, 1, -0.5, 0, 0.5, 1, -0.5, 0, 0.5, 1, 0, 0]
    ph = [0, 0, 0, 0, 0.25, 0.25, 0.25, 0.25, -0.25, -0.25, -0.25, -0.25, 0.5, -0.5]
    for near in (False, True):
        g = P().ProjectsOp(171, 256, th, ph, 0.5, near, 0, False).forward(x.to(DEV))[0]
        c = O.ProjectsOp(171, 256, th, ph, 0.5, near).forward(x)[0]
        assert g.shape == (14, 3, 171, 256)
        same(g, c)


def test_context_reshape_mask_gmm_loss():
    x = rnd(2, 42, 5, 7, seed=10)
    same(P().ContextReshapeOp(14, 0, False).forward(x.to(DEV))[0], O.ContextReshapeOp(14).forward(x)[0])
    for constrain in (1, 2, 5, 6):
        w = rnd(42, 42, 5, 5, seed=11) + 0.5
        wg = w.clone().to(DEV)
        P().MaskConstrainOp(constrain, 14, 0, False).forward(wg)
        wc = w.clone()
        O.MaskConstrainOp(constrain, 14).forward(wc)
        same(wg, wc)
    m = 257
    wt = torch.softmax(rnd(m, 3, seed=12), 1).contiguous()
    dl = rnd(m, 3, seed=13) * 3 + 0.05
    mu = rnd(m, 3, seed=14) * 8 - 3.5
    lb = torch.randint(0, 8, (m, 1), generator=torch.Generator().manual_seed(15)).float() - 3.5
    g = P().EntropyGmmOp(3, 0, 0, False).forward(wt.to(DEV), dl.to(DEV), mu.to(DEV), lb.to(DEV))[0]
    c = O.EntropyGmmOp(3, 0).forward(wt, dl, mu, lb)[0]
    assert (g.cpu() - c).abs().max().item() < 1e-5


def test_gmm_table_matches_oracle_and_is_a_cdf():
    n = 4000
    g = torch.Generator().manual_seed(16)
    raw = torch.randn(3, 3, 40, 100, generator=g) * 2   # sections: weights | deltas | means
    raw[1] = raw[1].abs() * 1.5 - 0.1                     # some negative deltas -> beta
    raw[2] = raw[2] * 2
    tnum = torch.tensor([n], dtype=torch.int32)
    dg = raw.clone().to(DEV)
    tg = P().EntropyGmmTableOp(8, 3.5, 3, 65536, 1e-6, 0, False).forward_batch(dg, tnum)[0]
    dc = raw.clone()
    tc = O.EntropyGmmTableOp(8, 3.5, 3, 65536, 1e-6).forward_batch(dc, tnum)[0]
    same(tg[:n], tc[:n].contiguous())
    same(dg.view(-1)[:n * 3], dc.view(-1)[:n * 3].contiguous())  # softmax written in place
    rows = tg[:n].cpu()
    assert (rows[:, 0] == 0).all() and (rows[:, 8] == 65536).all()
    assert (rows[:, 1:] - rows[:, :-1] >= 1).all()
    # libm erf instead of the published polynomial: at most one count apart, rarely
    O.set_detmath(False)
    tl = O.EntropyGmmTableOp(8, 3.5, 3, 65536, 1e-6).forward_batch(raw.clone(), tnum)[0][:n]
    diff = (rows - tl).abs()
    assert diff.max().item() <= 1 and (diff > 0).float().mean().item() < 0.02


def _wavefront_net(mod, nimg, h, w, ngroup, seed, steps=None):
    """drive DInput2 -> [pad -> conv(+act)] x2 -> add -> extract for all steps with
    backend `mod`; returns everything observable"""
    dev = DEV if mod is not O else "cpu"
    ctx = mod.EntropyContextOp(16, 18, W16, 0, False)
    ctx.start_context(w)
    addr = ctx.addr()
    g = torch.Generator().manual_seed(seed)
    w1 = (torch.randn(3, ngroup * 3, ngroup * 1, 5, 5, generator=g) * 0.2).to(dev)
    b1 = (torch.randn(3, ngroup * 3, generator=g) * 0.1).to(dev)
    a1 = (torch.rand(3, ngroup * 3, generator=g)).to(dev)
    w2 = (torch.randn(3, ngroup * 3, ngroup * 3, 5, 5, generator=g) * 0.1).to(dev)
    b2 = (torch.randn(3, ngroup * 3, generator=g) * 0.1).to(dev)
    a2 = (torch.rand(3, ngroup * 3, generator=g)).to(dev)
    w3 = (torch.randn(3, ngroup * 3, ngroup * 3, 5, 5, generator=g) * 0.1).to(dev)
    b3 = (torch.randn(3, ngroup * 3, generator=g) * 0.1).to(dev)
    data = torch.randint(0, 8, (nimg * 16, ngroup, h, w), generator=g).float().to(dev)
    fill = mod.PseudoFillOp(0, 16, 0, 0, addr, 2, 0, False)
    data = fill.forward(data)[0]
    ipt = mod.DInput2Op(ngroup, 16, 2, -3.5, 3, addr, 0, False)
    pad1 = mod.EntropyCtxPadRun2Op(2, 16, ngroup, True, addr, 0, False)
    conv1 = mod.EntropyConv2Op(16, ngroup, ngroup, ngroup * 3, 5, 5, 2, 2, addr, 0, False)
    pad2 = mod.EntropyCtxPadRun2Op(2, 16, ngroup, False, addr, 0, False)
    conv2 = mod.EntropyConv2Op(16, ngroup * 3, ngroup, ngroup * 3, 5, 6, 2, 2, addr, 0, False)
    add = mod.EntropyAddOp(16, ngroup * 3, ngroup, 2, addr, 0, False)
    pad3 = mod.EntropyCtxPadRun2Op(2, 16, ngroup, False, addr, 0, False)
    conv3 = mod.EntropyConv2Op(16, ngroup * 3, ngroup, ngroup * 3, 5, 6, 2, 0, addr, 0, False)
    ext = mod.DExtract2Op(16, ngroup, True, addr, 0, False)
    lab = mod.DExtract2Op(16, ngroup, True, addr, 0, False)
    gmm = mod.EntropyGmmTableOp(8, 3.5, 3, 65536, 1e-6, 0, False)
    label = torch.zeros((nimg, 1, h * 16, w), device=dev)
    nsteps = h * 16 + w + ngroup - 2 if steps is None else steps
    tables, counts = [], []
    for _ in range(nsteps):
        b = ipt.forward(label)[0]
        y1 = conv1.forward_act_batch(pad1.forward(b)[0], w1, b1, a1)[0]
        y2 = conv2.forward_act_batch(pad2.forward(y1)[0], w2, b2, a2)[0]
        y2 = add.forward(y2, y1)[0]
        y3 = conv3.forward_batch(pad3.forward(y2)[0], w3, b3)[0]
        z, le = ext.forward_batch(y3)
        vec = gmm.forward_batch(z, le)[0]
        n = int(le[0])
        label, _ = lab.forward(data)
        tables.append(vec[:n].detach().cpu().clone())
        counts.append(n)
    return dict(ctx=b.cpu(), y1=y1.cpu(), y2=y2.cpu(), y3=y3.cpu(), tables=tables, counts=counts, data=data.cpu())


@pytest.mark.parametrize("nimg,h,w,ngroup", [(1, 2, 64, 14), (2, 1, 64, 4)])
def test_entropy_wavefront_ops_bit_exact(nimg, h, w, ngroup):
    g = _wavefront_net(P(), nimg, h, w, ngroup, seed=21)
    c = _wavefront_net(O, nimg, h, w, ngroup, seed=21)
    assert g["counts"] == c["counts"]
    assert sum(g["counts"]) == int((g["data"].shape[1] * h * nimg) * sum(int(x) for x in O.widths_v3(W16, 16, 16 * h, w)))
    for k in ("ctx", "y1", "y2", "y3"):
        assert torch.equal(g[k], c[k]), k
    for tg, tc in zip(g["tables"], c["tables"]):
        assert torch.equal(tg, tc)
    # the scattered context equals the symbols minus 3.5 inside the valid region
    ctx = g["ctx"][:nimg * 16, :, 2:-2, 2:-2]
    widths = O.widths_v3(W16, 16, 16 * h, w)
    for t in range(16):
        v = int(widths[t])
        assert torch.equal(ctx[t::16, :, :, :v] + 3.5, g["data"][t::16, :, :, :v])


@pytest.mark.parametrize("cfg", [
    # tn, cin, h, w, cout, k, stride
    (2, 192, 6, 70, 192, 3, 1), (1, 96, 5, 66, 96, 3, 1), (2, 3, 9, 131, 192, 3, 2), (1, 192, 4, 64, 12, 3, 1),
    (2, 192, 4, 70, 96, 1, 1), (1, 96, 3, 64, 192, 1, 1), (2, 192, 5, 131, 192, 1, 2), (1, 192, 4, 66, 768, 3, 1),
    (1, 192, 7, 130, 192, 3, 2), (3, 20, 3, 40, 40, 3, 1),
])
def test_tile_conv_bit_exact_vs_fmaf_chain(cfg):
    tn, cin, h, w, cout, k, stride = cfg
    g = torch.Generator().manual_seed(31)
    x = torch.randn(tn, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (1.0 / np.sqrt(cin * k * k))
    b = torch.randn(cout, generator=g)
    sl = torch.rand(cout, generator=g)
    owner = type("Owner", (), {})()
    for slope in (None, sl):
        y = P().tile_conv2d(owner, x.to(DEV), wt.to(DEV), b.to(DEV), stride,
                            slope.to(DEV) if slope is not None else None)
        ref = O.conv2d_chain(x, wt, b, stride, slope)
        same(y, ref)
        t = O.tile_conv2d(None, x, wt, b, stride, slope)
        assert (y.cpu() - t).abs().max().item() < 1e-4


def test_tile_conv_dead_column_skip():
    g = torch.Generator().manual_seed(32)
    x = torch.randn(16, 8, 4, 258, generator=g)
    wt = torch.randn(32, 8, 3, 3, generator=g) * 0.1
    b = torch.randn(32, generator=g)
    limit = torch.tensor([70, 128, 129, 256, 0, 64, 1, 300] * 2, dtype=torch.int32)
    owner = type("Owner", (), {})()
    y = P().tile_conv2d(owner, x.to(DEV), wt.to(DEV), b.to(DEV), 1, None, limit.to(DEV), 16).cpu()
    ref = O.conv2d_chain(x, wt, b, 1, None)
    for t in range(16):
        first_dead = ((int(limit[t]) + 63) // 64) * 64 if limit[t] > 0 else 0
        assert torch.equal(y[t, :, :, :first_dead], ref[t, :, :, :first_dead])
        assert y[t, :, :, first_dead:].abs().max().item() == 0 if first_dead < 256 else True


@pytest.mark.gpu
@pytest.mark.parametrize("inverse", [False, True])
def test_gdn_fused_matches_reference_formula(hip_backend, inverse):
    """PseudoGDNV2 on the GPU (one fused launch) against the reference's op-by-op
    formula (PseudoContextV2.py:133-216) evaluated with torch on the same device"""
    import torch
    from pseudocylindrical_convolution_amd.PCONV_operator import PseudoContextV2, PseudoGDNV2
    torch.manual_seed(3)
    ctx = PseudoContextV2(16, True, device=0)
    gdn = PseudoGDNV2(192, 16, ctx, 0, inverse=inverse)
    with torch.no_grad():
        gdn.gamma.add_(torch.rand_like(gdn.gamma) * 0.02)
        gdn.beta.add_(torch.rand_like(gdn.beta) * 0.1)
    x = torch.randn(16, 192, 8, 256, device="cuda")
    ctx.setup_context(256)
    with torch.no_grad():
        fused = gdn(x)
    with torch.enable_grad():
        plain = gdn(x).detach()  # autograd on: the reference formula
    assert fused.shape == plain.shape
    assert torch.allclose(fused, plain, rtol=1e-5, atol=1e-6), (fused - plain).abs().max()
    # dead columns are zero, the valid ones are not
    w0 = int(ctx.produce_fill_param(0, 8, 256)[0])
    assert fused[0, :, :, w0:].abs().sum() == 0 and fused[0, :, :, :w0].abs().sum() > 0


@pytest.mark.parametrize("cfg", [(16, 192, 6, 130, 192, 3, 1), (16, 96, 4, 66, 192, 1, 1), (16, 192, 9, 131, 96, 3, 2)])
def test_tile_conv_epilogue_and_views(cfg):
    """the fused epilogue (sigmoid or PReLU, * gate, + residual, trim) is the oracle's
    fmaf-chain convolution followed by the same element-wise operations, bit for bit
    where no transcendental is involved; strided inputs / outputs (interior views of
    padded buffers, `ring`) give the same values as dense tensors"""
    tn, cin, h, w, cout, k, stride = cfg
    g = torch.Generator().manual_seed(41)
    x = torch.randn(tn, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (1.0 / np.sqrt(cin * k * k))
    b = torch.randn(cout, generator=g)
    sl = torch.rand(cout, generator=g)
    ho, wo = (h - k) // stride + 1, (w - k) // stride + 1
    res = torch.randn(tn, cout, ho, wo, generator=g)
    gate = torch.randn(tn, cout, ho, wo, generator=g)
    limit = torch.tensor([wo // 2, wo, wo - 3, 5] * 4, dtype=torch.int32)
    owner = type("Owner", (), {})()
    xg, wg, bg = x.to(DEV), wt.to(DEV), b.to(DEV)
    # PReLU + residual + trim: exact
    y = P().tile_conv2d(owner, xg, wg, bg, stride, sl.to(DEV), limit.to(DEV), 16, residual=res.to(DEV), trim=True)
    ref = res + O.conv2d_chain(x, wt, b, stride, sl)
    for t in range(tn):
        ref[t, :, :, int(limit[t % 16]):] = 0
    same(y, ref)
    # sigmoid + gate + residual: the exponential differs from torch's in the last bits
    y2 = P().tile_conv2d(owner, xg, wg, bg, stride, None, None, 0, sigmoid=True, gate=gate.to(DEV),
                         residual=res.to(DEV)).cpu()
    ref2 = res + gate * torch.sigmoid(O.conv2d_chain(x, wt, b, stride, None))
    assert (y2 - ref2).abs().max().item() < 2e-6
    # views: input, residual and gate inside padded buffers, output into a ring buffer
    def inside(t, p):
        buf = torch.full((t.shape[0], t.shape[1], t.shape[2] + 2 * p, t.shape[3] + 2 * p), float("nan"), device=DEV)
        buf[:, :, p:-p, p:-p] = t.to(DEV)
        return buf[:, :, p:-p, p:-p]
    y3 = P().tile_conv2d(owner, inside(x, 2), wg, bg, stride, sl.to(DEV), limit.to(DEV), 16,
                         residual=inside(res, 1), gate=inside(gate, 3), trim=True, ring=2)
    assert not y3.is_contiguous() and y3._pconv_ring[1] == 2
    y4 = P().tile_conv2d(owner, xg, wg, bg, stride, sl.to(DEV), limit.to(DEV), 16, residual=res.to(DEV),
                         gate=gate.to(DEV), trim=True)
    assert torch.equal(y3, y4)


@pytest.mark.parametrize("shape,pad", [((16, 4, 16, 512), 1), ((16, 3, 8, 256), 2), ((32, 2, 2, 64), 2),
                                       ((32, 2, 2, 64), 1)])
def test_pseudo_pad_ring_equals_full_pad(shape, pad):
    """PseudoPad of a tensor that already lives inside a padded buffer (only the ring is
    written) against the copying pad of the same values, which is pinned to the oracle above"""
    x = rnd(*shape, seed=5)
    gctx = P().PseudoContextOp(16, 20, W16, 0, False)
    xg = P().PseudoFillOp(0, 16, 0, 0, gctx.addr(), 0, 0, False).forward(x.clone().to(DEV))[0]
    full = P().PseudoPadOp(pad, 16, gctx.addr(), 0, False).forward(xg)[0].clone()
    store = 2
    buf = torch.full((shape[0], shape[1], shape[2] + 2 * store, shape[3] + 2 * store), float("nan"), device=DEV)
    view = buf[:, :, store:-store, store:-store]
    view.copy_(xg)
    view._pconv_ring = (buf, store)
    op = P().PseudoPadOp(pad, 16, gctx.addr(), 0, False)
    if pad < store:
        # a wider pad used the buffer before: its wrap columns must not survive
        P().PseudoPadOp(store, 16, gctx.addr(), 0, False).forward_ring(view)
    ring = op.forward_ring(view)
    assert ring.data_ptr() != full.data_ptr() and tuple(ring.shape) == tuple(full.shape)
    assert torch.equal(ring, full)


@pytest.mark.parametrize("cfg", [(16, 192, 6, 70, 768, 3), (16, 192, 4, 64, 768, 1), (2, 192, 6, 70, 12, 3)])
def test_tile_conv_depth_to_width_store(cfg):
    """conv + PReLU + DtowOp(2, True) with the pixel shuffle done by the convolution's store
    (dense and into a ring buffer) against the separate ops"""
    tn, cin, h, w, cout, k = cfg
    g = torch.Generator().manual_seed(43)
    x = torch.randn(tn, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (1.0 / np.sqrt(cin * k * k))
    b = torch.randn(cout, generator=g)
    sl = torch.rand(cout, generator=g)
    owner = type("Owner", (), {})()
    limit = torch.tensor([w - k + 1, 40, 64, 3] * 4, dtype=torch.int32).to(DEV)
    for slope in (None, sl.to(DEV)):
        plain = P().tile_conv2d(owner, x.to(DEV), wt.to(DEV), b.to(DEV), 1, slope, limit, 16)
        shuffled = P().DtowOp(2, True, 0, False).forward(plain)[0]
        for ring in (0, 2):
            fused = P().tile_conv2d(owner, x.to(DEV), wt.to(DEV), b.to(DEV), 1, slope, limit, 16, d2w=True, ring=ring)
            assert tuple(fused.shape) == tuple(shuffled.shape)
            assert torch.equal(fused, shuffled)


@pytest.mark.parametrize("cin,cout,shape", [(192, 96, (16, 34, 1026)), (96, 192, (16, 34, 1026)),
                                            (192, 192, (16, 34, 1026)), (192, 768, (16, 18, 514)),
                                            (192, 192, (16, 3, 70)), (96, 96, (2, 5, 7))])
def test_resident_1x1_equals_tiled_kernel(cin, cout, shape, monkeypatch):
    """the weight-resident, register-blocked 1x1 kernel (whole [K][BM] slab in LDS, 16-byte B
    loads straight from global memory, 16-byte stores, several row groups per workgroup)
    issues the same MFMA chain per output as the tiled kernel: bit-identical results, ragged
    edges (lanes past the edge recompute the last 4 columns), every epilogue, strided views"""
    tn, h, w = shape
    g = torch.Generator().manual_seed(47)
    x = torch.randn(tn, cin, h, w, generator=g).to(DEV)
    wt = (torch.randn(cout, cin, 1, 1, generator=g) * (1.0 / np.sqrt(cin))).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    sl = torch.rand(cout, generator=g).to(DEV)
    res = torch.randn(tn, cout, h, w, generator=g).to(DEV)
    gate = torch.randn(tn, cout, h, w, generator=g).to(DEV)
    limit = torch.tensor([w, (w * 7) // 8, w // 2 + 1, 64, 65, 1, w, 130] * 2, dtype=torch.int32).clamp(max=w).to(DEV)
    owner = type("Owner", (), {})()

    def variants():
        out = [P().tile_conv2d(owner, x, wt, b, 1, sl, limit, 16, residual=res, trim=True, ring=2).clone(),
               P().tile_conv2d(owner, x, wt, b, 1, None, None, 0, sigmoid=True, gate=gate, residual=res).clone(),
               P().tile_conv2d(owner, x, wt, None, 1, None, limit, 16).clone()]
        if cout % 4 == 0:
            out.append(P().tile_conv2d(owner, x, wt, b, 1, sl, limit, 16, d2w=True, ring=2).clone())
            out.append(P().tile_conv2d(owner, x, wt, b, 1, None, None, 0, d2w=True).clone())
        if cin == cout:
            gam = (torch.rand(cin, cin, generator=torch.Generator().manual_seed(5)) * 0.01 + torch.eye(cin) * 0.1).to(DEV)
            beta = (torch.rand(cin, generator=torch.Generator().manual_seed(6)) + 0.5).to(DEV)
            for inverse in (False, True):
                out.append(P().tile_gdn(owner, x, gam, beta, inverse, limit, 16, res, 2).clone())
        return out

    monkeypatch.setenv("PCONV_CONV1X1", "tiled")
    tiled = variants()   # (the default way out: 16-byte quads through the stage memory for full tiles)
    # the tiled kernel's element-wise ways out -- rows requested a batch ahead as straight-line code ("pipe") and
    # conv_epilogue's whole-tile batches ("batch"): identical bits
    monkeypatch.setenv("PCONV_CONV1X1_WAYOUT", "batch")
    batched = variants()
    monkeypatch.setenv("PCONV_CONV1X1_WAYOUT", "pipe")
    piped = variants()
    monkeypatch.delenv("PCONV_CONV1X1_WAYOUT")
    for i, (a, r) in enumerate(zip(tiled, piped)):
        assert torch.equal(a, r), "pipelined way out, variant %d: max abs diff %g" % (i, (a - r).abs().max().item())
    for i, (a, r) in enumerate(zip(tiled, batched)):
        assert torch.equal(a, r), "way out, variant %d: max abs diff %g" % (i, (a - r).abs().max().item())
    monkeypatch.setenv("PCONV_CONV1X1", "resident")
    resident = variants()
    monkeypatch.delenv("PCONV_CONV1X1")
    for i, (a, r) in enumerate(zip(tiled, resident)):
        assert torch.isfinite(r).all()
        assert torch.equal(a, r), "variant %d: max abs diff %g" % (i, (a - r).abs().max().item())
    # and against the oracle's fmaf chain on a slice of the batch
    ref = O.conv2d_chain(x[:1].cpu(), wt.cpu(), None, 1, None)
    assert torch.equal(resident[2][:1].cpu(), ref)   # tile 0 is live over its whole width


@pytest.mark.parametrize("cin,cout,shape", [(192, 96, (16, 34, 1026)), (96, 192, (16, 34, 1026)),
                                            (192, 192, (16, 34, 1026)), (192, 768, (16, 18, 514)),
                                            (192, 192, (16, 3, 70)), (96, 96, (2, 5, 7)), (96, 192, (128, 8, 256)),
                                            (192, 192, (1, 64, 2048))])
def test_quad_way_out_equals_elementwise_way_out(cin, cout, shape, monkeypatch):
    """full tiles of the 1x1 / GDN layers leave through the stage memory in 16-byte quads (conv_epilogue_quads, the
    default): the same epilogue operations per output as the element-wise, pipelined way out (PCONV_CONV1X1_WAYOUT=
    pipe) -- identical bits: ragged right / lower edges (element-wise fallback), dead tiles, one tile per workgroup
    and many, both cout block widths, residual, PReLU, trim, ring-buffer views, the GDN pair"""
    tn, h, w = shape
    g = torch.Generator().manual_seed(53)
    x = torch.randn(tn, cin, h, w, generator=g).to(DEV)
    wt = (torch.randn(cout, cin, 1, 1, generator=g) * (1.0 / np.sqrt(cin))).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    sl = torch.rand(cout, generator=g).to(DEV)
    res = torch.randn(tn, cout, h, w, generator=g).to(DEV)
    limit = torch.tensor([w, (w * 7) // 8, w // 2 + 1, 64, 65, 1, w, 130] * 2, dtype=torch.int32).clamp(max=w).to(DEV)
    owner = type("Owner", (), {})()

    def variants():
        out = [P().tile_conv2d(owner, x, wt, b, 1, sl, limit, 16, residual=res, trim=True, ring=2).clone(),
               P().tile_conv2d(owner, x, wt, b, 1, sl, None, 0, residual=res).clone(),
               P().tile_conv2d(owner, x, wt, None, 1, None, limit, 16).clone(),
               P().tile_conv2d(owner, x, wt, b, 1, sl, limit, 16, trim=True).clone()]
        if cin == cout:
            gam = (torch.rand(cin, cin, generator=torch.Generator().manual_seed(5)) * 0.01 + torch.eye(cin) * 0.1).to(DEV)
            beta = (torch.rand(cin, generator=torch.Generator().manual_seed(6)) + 0.5).to(DEV)
            for inverse in (False, True):
                out.append(P().tile_gdn(owner, x, gam, beta, inverse, limit, 16, res, 2).clone())
                out.append(P().tile_gdn(owner, x, gam, beta, inverse, limit, 16, None, 0).clone())
        return out

    monkeypatch.setenv("PCONV_CONV1X1_WAYOUT", "pipe")   # the element-wise, pipelined way out
    piped = variants()
    monkeypatch.delenv("PCONV_CONV1X1_WAYOUT")
    quads = variants()                                    # the default
    for i, (a, r) in enumerate(zip(piped, quads)):
        assert torch.isfinite(r).all()
        assert torch.equal(a, r), "variant %d: max abs diff %g" % (i, (a - r).abs().max().item())


@pytest.mark.parametrize("cin,cout,shape", [(192, 192, (16, 33, 1025)), (192, 96, (16, 17, 257)), (3, 192, (2, 65, 131))])
def test_stride2_quad_way_out_equals_elementwise(cin, cout, shape, monkeypatch):
    """the stride-2 3x3 layers (Down blocks) share the 1x1 layers' workgroup tiles and their quad way out through
    the stage memory: identical bits to the element-wise way out (PCONV_CONV1X1_WAYOUT=batch), with PReLU, residual,
    trim, dead tiles, ragged edges and ring-buffer outputs"""
    tn, h, w = shape
    g = torch.Generator().manual_seed(59)
    x = torch.randn(tn, cin, h, w, generator=g).to(DEV)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * (1.0 / np.sqrt(cin * 9))).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    sl = torch.rand(cout, generator=g).to(DEV)
    ho, wo = (h - 3) // 2 + 1, (w - 3) // 2 + 1
    res = torch.randn(tn, cout, ho, wo, generator=g).to(DEV)
    limit = torch.tensor([wo, (wo * 7) // 8, wo // 2 + 1, 64, 65, 1, wo, 130] * 2, dtype=torch.int32).clamp(max=wo)[:16].to(DEV)
    owner = type("Owner", (), {})()

    w1 = wt[:, :, 1:2, 1:2].contiguous()                 # and the 1x1 stride-2 shortcut of the same blocks
    x1 = x[:, :, :2 * ho - 1, :2 * wo - 1].contiguous()  # (same output geometry)

    def variants():
        return [P().tile_conv2d(owner, x, wt, b, 2, sl, limit, 16, residual=res, trim=True, ring=1).clone(),
                P().tile_conv2d(owner, x, wt, b, 2, sl, None, 0).clone(),
                P().tile_conv2d(owner, x, wt, None, 2, None, limit, 16, residual=res).clone(),
                P().tile_conv2d(owner, x1, w1, b, 2, sl, limit, 16, residual=res, trim=True, ring=1).clone(),
                P().tile_conv2d(owner, x1, w1, None, 2, None, None, 0).clone()]

    quads = variants()
    monkeypatch.setenv("PCONV_CONV1X1_WAYOUT", "batch")
    plain = variants()
    monkeypatch.delenv("PCONV_CONV1X1_WAYOUT")
    for i, (a, r) in enumerate(zip(plain, quads)):
        assert torch.isfinite(r).all()
        assert torch.equal(a, r), "variant %d: max abs diff %g" % (i, (a - r).abs().max().item())
    ref = O.conv2d_chain(x[:1].cpu(), wt.cpu(), None, 2, None)
    assert torch.equal(P().tile_conv2d(owner, x[:1], wt, None, 2, None, None, 0).cpu(), ref)


@pytest.mark.parametrize("cin,cout,shape", [(192, 12, (16, 18, 130)), (24, 16, (2, 11, 70)), (6, 3, (3, 5, 9)),
                                            (192, 12, (1, 66, 2050))])
def test_small_cout_3x3_kernel(cin, cout, shape, monkeypatch):
    """3x3 stride-1 layers with <= 16 couts (the codec's 192 -> 12 output layer) run on v_mfma_f32_16x16x4_f32 tiles
    of 16 couts: the same k-ascending fmaf chain per output as the 32-cout tile of conv_mfma_kernel
    (PCONV_CONV_SMALL=0) and as the oracle -- bias, PReLU, trim, dead tiles, ragged edges and channel counts, the
    depth-to-width store, ring-buffer outputs"""
    tn, h, w = shape
    g = torch.Generator().manual_seed(61)
    x = torch.randn(tn, cin, h, w, generator=g).to(DEV)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * (1.0 / np.sqrt(cin * 9))).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    sl = torch.rand(cout, generator=g).to(DEV)
    wo = w - 2
    npart = 16 if tn % 16 == 0 else 1
    limit = torch.tensor([wo, (wo * 7) // 8, wo // 2 + 1, 64, 65, 1, wo, 130] * 2, dtype=torch.int32).clamp(max=wo)[:npart].to(DEV)
    owner = type("Owner", (), {})()

    def variants():
        out = [P().tile_conv2d(owner, x, wt, b, 1, sl, limit, npart, trim=True, ring=1).clone(),
               P().tile_conv2d(owner, x, wt, None, 1, None, None, 0).clone(),
               P().tile_conv2d(owner, x, wt, b, 1, None, limit, npart).clone()]
        if cout % 4 == 0:
            out.append(P().tile_conv2d(owner, x, wt, b, 1, None, limit, npart, d2w=True).clone())
            out.append(P().tile_conv2d(owner, x, wt, b, 1, sl, None, 0, d2w=True, ring=2).clone())
        return out

    small = variants()
    monkeypatch.setenv("PCONV_CONV_SMALL", "0")
    tiled = variants()
    monkeypatch.delenv("PCONV_CONV_SMALL")
    for i, (a, r) in enumerate(zip(tiled, small)):
        assert torch.isfinite(r).all()
        assert torch.equal(a, r), "variant %d: max abs diff %g" % (i, (a - r).abs().max().item())
    ref = O.conv2d_chain(x[:1].cpu(), wt.cpu(), b.cpu(), 1, sl.cpu())
    assert torch.equal(P().tile_conv2d(owner, x[:1], wt, b, 1, sl, None, 0).cpu(), ref)


def test_timeit_prints_like_the_reference_timer(capsys):
    """`timeit=True` (last constructor argument of every op, base_opt.hpp:7-8, timer.h:32-44): events
    around the op's call, `<head> Elapsed time : <ms> ms` on stdout"""
    x = rnd(16, 8, 4, 64).to(DEV)
    op = P().DtowOp(2, True, 0, True)
    out = op.forward(x)[0]
    assert tuple(out.shape) == (16, 2, 8, 128)
    printed = capsys.readouterr().out
    assert "DtowOp.forward Elapsed time : " in printed and printed.strip().endswith("ms")
    quiet = P().DtowOp(2, True, 0, False)
    quiet.forward(x)
    assert capsys.readouterr().out == ""


def test_hbm_probe_records_the_gather_kernels(monkeypatch):
    """bench.py's `hbm` rows: every launch of the HBM-bound gather / permute ops is bracketed by events
    with the algorithmic bytes of SURVEY 8d"""
    rec = type("Probe", (), {"records": []})()
    monkeypatch.setattr(P(), "hbm_probe", rec)
    x = rnd(1, 3, 256, 512).to(DEV)
    s = P().SphereSliceOp(16, 0, 0, W16, 0, False).forward(x)[0]
    P().SphereUsliceOp(16, 0, 0, W16, 0, False).forward(s)
    P().DtowOp(2, True, 0, False).forward(rnd(16, 8, 4, 64).to(DEV))
    torch.cuda.synchronize()
    kernels = [r[0] for r in rec.records]
    assert kernels == ["slice_kernel", "uslice_kernel", "dtow2_kernel"]
    assert rec.records[0][2] == 8.0 * x.numel() and all(r[3].elapsed_time(r[4]) > 0 for r in rec.records)


def test_leaky_clip_kernel_equals_the_reference_formulation(hip_backend):
    """ClipData.forward (model_zoo_v2.py:8-26): the one-pass in-place kernel against the reference's masked index
    assignments, bit for bit (every operation rounds on its own), odd lengths and unaligned views included; the
    module takes the kernel under no_grad and keeps the autograd formulation otherwise"""
    from pseudocylindrical_convolution_amd.model_zoo_v2 import ClipData
    g = torch.Generator().manual_seed(21)
    for n, off in ((1 << 20, 0), (1000003, 0), (4099, 1), (7, 3)):
        base = ((torch.rand(n + 8, generator=g) - 0.25) * 2.0)
        base[:6] = torch.tensor([0.0, 1.0, -0.0, 1.0000001, -1e-30, float("inf")])
        x = base.cuda()[off:off + n]
        ref = x.clone()
        below, above = x < 0, x > 1
        ref[below] = x[below] * 0.01
        ref[above] = 1 + (x[above] - 1) * 0.01
        assert (x.data_ptr() % 16 == 0) == (off == 0)       # off > 0: an unaligned start, the scalar path
        got = hip_backend.leaky_clip_(x)                     # in place (x is a contiguous view)
        assert got.data_ptr() == x.data_ptr() and torch.equal(got, ref)
    x = ((torch.rand(3, 5, 7, generator=g) - 0.25) * 2.0).cuda()
    with torch.no_grad():
        y = ClipData()(x)
    ref = x.clone()
    ref[x < 0] = x[x < 0] * 0.01
    ref[x > 1] = 1 + (x[x > 1] - 1) * 0.01
    assert torch.equal(y, ref) and y.data_ptr() != x.data_ptr()
    xg = x.clone().requires_grad_(True)
    yg = ClipData()(xg)
    assert torch.equal(yg.detach(), ref)
    yg.sum().backward()
    assert torch.equal(xg.grad, torch.where((x < 0) | (x > 1), torch.full_like(x, 0.01), torch.ones_like(x)))
