"""Training path, CPU: the oracle's backward restatements of the linear
geometry ops are the exact transposes of its forward restatements -- <A x, g> == <x, A^T g>
for random x and g -- and agree with torch autograd through an equivalent dense formulation
where one exists.  (The reference cannot run here and its own backward sums in an undefined
order; the adjoint identity is what pins these restatements.)"""
import numpy as np
import pytest
import torch

from oracle import pconv_cpu as O

W16 = [15., 31., 54., 63., 63., 64., 64., 64., 64., 64., 64., 63., 63., 54., 31., 15.]


def _adjoint(fwd, bwd, x, g_shape, seed=0, rtol=2e-5):
    gen = torch.Generator().manual_seed(seed)
    y = fwd(x).clone()
    assert tuple(y.shape) == tuple(g_shape)
    g = torch.randn(g_shape, generator=gen)
    gx = bwd(g).clone()
    assert gx.shape == x.shape
    lhs = (y.double() * g.double()).sum().item()
    rhs = (x.double() * gx.double()).sum().item()
    assert abs(lhs - rhs) <= rtol * max(abs(lhs), abs(rhs), 1.0), (lhs, rhs)
    return gx


@pytest.mark.parametrize("pad", [0, 1])
def test_slice_backward_is_the_transpose(pad):
    op = O.SphereSliceOp(16, 0, pad, W16)
    x = torch.randn(2, 3, 64, 128, generator=torch.Generator().manual_seed(1))
    gx = _adjoint(lambda t: op.forward(t)[0], lambda g: op.backward(g)[0], x, (32, 3, 4 + 2 * pad, 128 + 2 * pad))
    assert gx.abs().sum() > 0


@pytest.mark.parametrize("pad", [0, 2])
def test_uslice_backward_is_the_transpose(pad):
    op = O.SphereUsliceOp(16, 0, pad, W16)
    x = torch.randn(32, 2, 4 + 2 * pad, 128 + 2 * pad, generator=torch.Generator().manual_seed(2))
    gx = _adjoint(lambda t: op.forward(t)[0], lambda g: op.backward(g)[0], x, (2, 2, 64, 128))
    # no gradient outside the valid interior
    widths = O.widths_v3(W16, 16, 64, 128)
    inner = gx[:, :, pad:gx.shape[2] - pad, pad:gx.shape[3] - pad] if pad else gx
    for t in range(16):
        assert inner[t::16, :, :, int(widths[t]):].abs().max().item() == 0 if widths[t] < 128 else True
    if pad:
        assert gx[:, :, :pad].abs().max().item() == 0 and gx[:, :, :, :pad].abs().max().item() == 0


@pytest.mark.parametrize("shape,pad", [((32, 2, 4, 128), 1), ((16, 3, 2, 64), 2), ((16, 1, 8, 256), 2)])
def test_pad_backward_is_the_transpose(shape, pad):
    ctx = O.PseudoContextOp(16, 20, W16)
    op = O.PseudoPadOp(pad, 16, ctx.addr())
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(3))
    # the forward ignores the dead columns, so their gradient is zero: compare on zeroed input
    xz = O.PseudoFillOp(0, 16, 0, 0, ctx.addr(), 0).forward(x.clone())[0]
    gshape = (shape[0], shape[1], shape[2] + 2 * pad, shape[3] + 2 * pad)
    gx = _adjoint(lambda t: op.forward(t)[0], lambda g: op.backward(g)[0], xz, gshape)
    widths = O.widths_v3(W16, 16, 16 * shape[2], shape[3])
    for t in range(16):
        assert gx[t::16, :, :, int(widths[t]):].abs().max().item() == 0 if widths[t] < shape[3] else True
    # the argument is not modified (the reference folds the wrap columns in place)
    g = torch.randn(gshape, generator=torch.Generator().manual_seed(4))
    g0 = g.clone()
    op.backward(g)
    assert torch.equal(g, g0)


def test_dtow_and_context_reshape_backward_invert_the_permutation():
    x = torch.randn(3, 36, 4, 6, generator=torch.Generator().manual_seed(5))
    for stride, d2w, xin in ((2, True, x), (3, True, x), (2, False, x[:, :9]), (2, False, x[:, :4])):
        op = O.DtowOp(stride, d2w)
        xin = xin.contiguous()
        y = op.forward(xin)[0].clone()
        assert torch.equal(op.backward(y)[0], xin)       # A^T A = I for a permutation
    cr = O.ContextReshapeOp(4)
    y = cr.forward(x)[0].clone()
    assert torch.equal(cr.backward(y)[0], x)
    assert torch.equal(y, x.view(3, 4, 9, 4, 6).permute(0, 1, 3, 4, 2).reshape(-1, 9))


def test_gmm_loss_backward_matches_autograd_of_the_formula():
    """EntropyGmm: -log(sum_k w_k (Phi(b_k) - Phi(a_k)) + 1e-7) (entropy_gmm_cuda.cu:36-127)"""
    O.set_detmath(False)
    m = 300
    g = torch.Generator().manual_seed(6)
    wt = torch.softmax(torch.randn(m, 3, generator=g), 1).double().requires_grad_()
    dl = (torch.rand(m, 3, generator=g) * 2 + 0.3).double().requires_grad_()
    mu = (torch.randn(m, 3, generator=g) * 2).double().requires_grad_()
    lb = (torch.randint(0, 8, (m, 1), generator=g).double() - 3.5).requires_grad_()
    cdf = lambda v: 0.5 + 0.5 * torch.erf(v / np.sqrt(2.0))
    p = (wt * (cdf((lb + 0.5 - mu) / dl) - cdf((lb - 0.5 - mu) / dl))).sum(1)
    loss = -torch.log(p + 1e-7)
    top = torch.randn(m, generator=g).double()
    (loss * top).sum().backward()
    op = O.EntropyGmmOp(3, 0)
    out = op.forward(wt.detach().float().contiguous(), dl.detach().float().contiguous(),
                     mu.detach().float().contiguous(), lb.detach().float().contiguous())[0]
    assert (out.double() - loss.detach()).abs().max().item() < 2e-3   # fp32 erf differences, amplified by -log
    dw, dd, dm, dlab = op.backward(top.float())
    for got, ref in ((dw, wt.grad), (dd, dl.grad), (dm, mu.grad), (dlab, lb.grad)):
        assert (got.double() - ref).abs().max().item() < 2e-3 * max(1.0, ref.abs().max().item())


def test_quantiser_autograd_bridge_on_the_oracle(oracle_backend):
    """PseudoQUANTV2 under autograd: loss.backward() fills x.grad (straight-through inside the
    valid columns), weight.grad (level table) and count.grad (the per-call histogram, as the
    reference returns it: PseudoContextV2.py:233-239)"""
    from pseudocylindrical_convolution_amd.PCONV_operator import PseudoContextV2, PseudoQUANTV2
    ctx = PseudoContextV2(16, True, device=0)
    q = PseudoQUANTV2(8, 8, 16, ctx, ntop=2).eval()
    x = (torch.rand(16, 8, 2, 64, generator=torch.Generator().manual_seed(11)) * 1.2 - 0.1).requires_grad_()
    val, idx = q(x)
    wgt = torch.rand(val.shape, generator=torch.Generator().manual_seed(12))
    (val * wgt).sum().backward()
    widths = O.widths_v3(W16, 16, 32, 64)
    for t in range(16):
        v = int(widths[t])
        assert torch.equal(x.grad[t::16, :, :, :v], wgt[t::16, :, :, :v])      # idx got no gradient
        assert x.grad[t::16, :, :, v:].abs().sum().item() == 0
    assert q.weight.grad is not None and q.weight.grad.abs().sum().item() > 0
    assert int(-q.count.grad.sum().item()) == 8 * 2 * 836                  # one count per valid element


@pytest.mark.parametrize("near", [False, True])
def test_projects_backward_is_the_transpose(near):
    th = [-0.5, 0, 0.5, 1, -0.5, 0, 0.5, 1, -0.5, 0, 0.5, 1, 0, 0]
    ph = [0, 0, 0, 0, 0.25, 0.25, 0.25, 0.25, -0.25, -0.25, -0.25, -0.25, 0.5, -0.5]
    op = O.ProjectsOp(19, 28, th, ph, 0.5, near)
    x = torch.randn(2, 3, 64, 128, generator=torch.Generator().manual_seed(21))
    _adjoint(lambda t: op.forward(t)[0], lambda g: op.backward(g)[0], x, (28, 3, 19, 28))
    cnt = op.backward(torch.ones(28, 3, 19, 28))[1]
    assert abs(cnt.sum().item() - 28 * 3 * 19 * 28) < 1.0     # the bilinear weights of a sample sum to 1


@pytest.mark.parametrize("shape,pad,version", [((32, 2, 4, 128), 2, 1), ((16, 3, 2, 64), 2, 1), ((16, 1, 8, 256), 1, 1),
                                               ((16, 2, 4, 128), 2, 0)])
def test_entropy_pad_is_causal_and_its_backward_is_the_transpose(shape, pad, version):
    """PseudoEntropyPadOp (pseudo_entropy_pad_cuda.cu): halo column j of a tile reads only source
    columns <= j (version 1), the left halo columns and the pole rows are zero, the first `pad`
    columns re-appear after the last valid one, and backward is the exact transpose"""
    ctx = O.PseudoEntropyContextOp(16, 20, version, W16)
    op = O.PseudoEntropyPadOp(pad, 16, ctx.addr())
    tn, c, h, w = shape
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(7))
    xz = O.PseudoFillOp(0, 16, 0, 0, ctx.addr(), 1).forward(x.clone())[0]
    gshape = (tn, c, h + 2 * pad, w + 2 * pad)
    gx = _adjoint(lambda t: op.forward(t)[0], lambda g: op.backward(g)[0], xz, gshape)
    widths = O.widths_v3(W16, 16, 16 * h, w)
    y = op.forward(xz)[0].clone()
    assert y[:, :, :, :pad].abs().max().item() == 0
    assert y[0::16, :, :pad].abs().max().item() == 0 and y[15::16, :, h + pad:].abs().max().item() == 0
    for t in range(16):
        v = int(widths[t])
        assert torch.equal(y[t::16, :, pad:pad + h, pad:pad + v], xz[t::16, :, :, :v])
        assert torch.equal(y[t::16, :, :, pad + v:2 * pad + v], y[t::16, :, :, pad:2 * pad])
        assert y[t::16, :, :, 2 * pad + v:].abs().sum().item() == 0
        if v < w:
            assert gx[t::16, :, :, v:].abs().max().item() == 0
    if version == 1:
        # causality: perturbing source column k changes no halo value left of column k of the reader
        k = 5
        xp = xz.clone()
        xp[:, :, :, k:] += 1.0
        xp = O.PseudoFillOp(0, 16, 0, 0, ctx.addr(), 1).forward(xp)[0]
        yp = op.forward(xp)[0]
        assert torch.equal(yp[:, :, :pad, pad:pad + k], y[:, :, :pad, pad:pad + k])
        assert torch.equal(yp[:, :, h + pad:, pad:pad + k], y[:, :, h + pad:, pad:pad + k])
    g = torch.randn(gshape, generator=torch.Generator().manual_seed(8))
    g0 = g.clone()
    op.backward(g)
    assert torch.equal(g, g0)
