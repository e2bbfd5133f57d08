"""GPU: backward kernels of the linear geometry ops against the oracle's restatements
(same inputs; the sums run in a different order -- LDS atomics / a fixed gather order here,
sequential scatter there -- so the comparison is <= 1e-5 relative, not bit for bit), the
adjoint identity on the GPU path itself, and autograd through the nn.Module wrappers."""
import numpy as np
import pytest
import torch

from oracle import pconv_cpu as O

pytestmark = pytest.mark.gpu

W16 = [15., 31., 54., 63., 63., 64., 64., 64., 64., 64., 64., 63., 63., 54., 31., 15.]
DEV = "cuda:0"


def P():
    from pseudocylindrical_convolution_amd import PCONV
    return PCONV


def close(a_gpu, b_cpu, tol=1e-5):
    a = a_gpu.detach().cpu()
    assert a.shape == b_cpu.shape
    scale = max(1.0, b_cpu.abs().max().item())
    assert (a - b_cpu).abs().max().item() <= tol * scale, (a - b_cpu).abs().max().item()


@pytest.mark.parametrize("shape,pad", [((1, 3, 512, 1024), 0), ((2, 2, 256, 512), 1)])
def test_slice_backward(shape, pad):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(*shape, generator=g)
    gop, cop = P().SphereSliceOp(16, 0, pad, W16, 0, False), O.SphereSliceOp(16, 0, pad, W16)
    yg, yc = gop.forward(x.to(DEV))[0], cop.forward(x)[0]
    grad = torch.randn(yc.shape, generator=g)
    gg, gc = gop.backward(grad.to(DEV))[0], cop.backward(grad)[0]
    close(gg, gc)
    # adjoint identity of the GPU pair
    lhs = (yg.double() * grad.to(DEV).double()).sum().item()
    rhs = (x.to(DEV).double() * gg.double()).sum().item()
    assert abs(lhs - rhs) <= 2e-5 * max(abs(lhs), abs(rhs), 1.0)


@pytest.mark.parametrize("shape,pad", [((16, 3, 32, 1024), 0), ((32, 2, 18, 516), 2)])
def test_uslice_backward(shape, pad):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(*shape, generator=g)
    gop, cop = P().SphereUsliceOp(16, 0, pad, W16, 0, False), O.SphereUsliceOp(16, 0, pad, W16)
    yg, yc = gop.forward(x.to(DEV))[0], cop.forward(x)[0]
    grad = torch.randn(yc.shape, generator=g)
    gg, gc = gop.backward(grad.to(DEV))[0], cop.backward(grad)[0]
    close(gg, gc)
    lhs = (yg.double() * grad.to(DEV).double()).sum().item()
    # forward reads only the valid interior: compare against the interior part of x
    rhs = (x.to(DEV).double() * gg.double()).sum().item()
    assert abs(lhs - rhs) <= 2e-5 * max(abs(lhs), abs(rhs), 1.0)


@pytest.mark.parametrize("shape,pad", [((16, 4, 16, 512), 1), ((32, 3, 8, 256), 2), ((16, 2, 2, 64), 2)])
def test_pad_backward(shape, pad):
    g = torch.Generator().manual_seed(3)
    gctx, octx = P().PseudoContextOp(16, 20, W16, 0, False), O.PseudoContextOp(16, 20, W16)
    gop, cop = P().PseudoPadOp(pad, 16, gctx.addr(), 0, False), O.PseudoPadOp(pad, 16, octx.addr())
    grad = torch.randn(shape[0], shape[1], shape[2] + 2 * pad, shape[3] + 2 * pad, generator=g)
    gd = grad.to(DEV)
    gg, gc = gop.backward(gd)[0], cop.backward(grad)[0]
    close(gg, gc)
    assert torch.equal(gd.cpu(), grad)                 # the argument is left alone
    x = O.PseudoFillOp(0, 16, 0, 0, octx.addr(), 0).forward(torch.randn(*shape, generator=g))[0]
    yg = gop.forward(x.to(DEV))[0]
    lhs = (yg.double() * gd.double()).sum().item()
    rhs = (x.to(DEV).double() * gg.double()).sum().item()
    assert abs(lhs - rhs) <= 2e-5 * max(abs(lhs), abs(rhs), 1.0)


@pytest.mark.parametrize("shape,pad,version", [((16, 4, 16, 512), 2, 1), ((32, 3, 8, 256), 2, 1), ((16, 2, 2, 64), 2, 1),
                                               ((16, 2, 4, 128), 1, 0), ((16, 9, 64, 1024), 2, 1)])
def test_entropy_pad_forward_backward(shape, pad, version):
    """PseudoEntropyPadOp: forward bit-exact (one fmul/fmul/fadd per halo value, no contraction on
    either side), backward within 1e-5, and <A x, g> == <x, A^T g> on the device"""
    g = torch.Generator().manual_seed(13)
    gctx, octx = P().PseudoEntropyContextOp(16, 20, version, W16, 0, False), O.PseudoEntropyContextOp(16, 20, version, W16)
    gop, cop = P().PseudoEntropyPadOp(pad, 16, gctx.addr(), 0, False), O.PseudoEntropyPadOp(pad, 16, octx.addr())
    x = torch.randn(*shape, generator=g)                # dead columns left dirty: both sides ignore them
    yg, yc = gop.forward(x.to(DEV))[0], cop.forward(x)[0]
    close(yg, yc, 1e-6)
    grad = torch.randn(yc.shape, generator=g)
    gd = grad.to(DEV)
    gg, gc = gop.backward(gd)[0], cop.backward(grad)[0]
    close(gg, gc)
    assert torch.equal(gd.cpu(), grad)
    xz = O.PseudoFillOp(0, 16, 0, 0, octx.addr(), 1).forward(x.clone())[0].to(DEV)
    yz = gop.forward(xz)[0]
    lhs = (yz.double() * gd.double()).sum().item()
    rhs = (xz.double() * gg.double()).sum().item()
    assert abs(lhs - rhs) <= 2e-5 * max(abs(lhs), abs(rhs), 1.0)


def test_context_reshape_dtow_gmm_backward():
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 42, 5, 7, generator=g)
    gop, cop = P().ContextReshapeOp(14, 0, False), O.ContextReshapeOp(14)
    yg, yc = gop.forward(x.to(DEV))[0], cop.forward(x)[0]
    grad = torch.randn(yc.shape, generator=g)
    assert torch.equal(gop.backward(grad.to(DEV))[0].cpu(), cop.backward(grad)[0])
    assert torch.equal(gop.backward(yg.clone())[0].cpu(), x)
    d = P().DtowOp(2, True, 0, False)
    y = d.forward(x[:, :36].contiguous().to(DEV))[0].clone()
    assert torch.equal(d.backward(y)[0].cpu(), x[:, :36])
    O.set_detmath(True)
    m = 257
    wt = torch.softmax(torch.randn(m, 3, generator=g), 1).contiguous()
    dl = torch.rand(m, 3, generator=g) * 3 + 0.05
    mu = torch.rand(m, 3, generator=g) * 8 - 3.5
    lb = torch.randint(0, 8, (m, 1), generator=g).float() - 3.5
    top = torch.randn(m, generator=g)
    gg = P().EntropyGmmOp(3, 0, 0, False)
    cg = O.EntropyGmmOp(3, 0)
    gg.forward(wt.to(DEV), dl.to(DEV), mu.to(DEV), lb.to(DEV))
    cg.forward(wt, dl, mu, lb)
    for a, b in zip(gg.backward(top.to(DEV)), cg.backward(top)):
        close(a, b, 1e-4)


def test_autograd_through_the_modules(hip_backend):
    """slice -> pad -> uslice as nn.Modules: loss.backward() reaches the input through the
    HIP backward kernels and matches the oracle-backend graph"""
    from pseudocylindrical_convolution_amd.PCONV_operator import (PseudoContextV2, PseudoPadV2, SphereSlice,
                                                                  SphereUslice, backend)
    from oracle import coder_cpu

    def run(dev):
        ctx = PseudoContextV2(16, True, device=0)
        net = [SphereSlice(16, pad=0, opt=True, device=0), PseudoPadV2(2, 16, ctx, device=0),
               SphereUslice(16, pad=2, opt=True, device=0)]
        x = torch.rand(1, 2, 256, 512, generator=torch.Generator().manual_seed(7)).to(dev).requires_grad_()
        wgt = torch.rand(1, 2, 256, 512, generator=torch.Generator().manual_seed(8)).to(dev)
        y = x
        for m in net:
            y = m(y)
        (y * wgt).sum().backward()
        return y.detach().cpu(), x.grad.detach().cpu()

    yg, gg = run(DEV)
    backend.use(O, coder_cpu)
    try:
        yc, gc = run("cpu")
    finally:
        backend.reset()
    assert torch.equal(yg, yc)
    assert gg.abs().sum() > 0
    assert (gg - gc).abs().max().item() <= 1e-5 * max(1.0, gc.abs().max().item())


@pytest.mark.parametrize("ntop", [1, 2])
def test_quant_backward(ntop):
    """PseudoQuantOp.backward (pseudo_quant_cuda.cu:197-311): straight-through value gradient plus the
    index gradient scaled by the local level width; level-table gradient = quantisation error summed
    into the levels at or below each element's level (float atomics here: 1e-4 relative)"""
    O.set_detmath(True)
    g = torch.Generator().manual_seed(9)
    x = torch.rand(16, 192, 2, 64, generator=g) * 1.2 - 0.1
    weight = torch.zeros(192, 8)
    weight[:, 0] = 1. / 9
    weight[:, 1:] = float(np.log(1. / 9))
    weight += torch.rand(192, 8, generator=g) * 0.05
    count = torch.zeros(192, 8)
    gctx, octx = P().PseudoContextOp(16, 20, W16, 0, False), O.PseudoContextOp(16, 20, W16)
    gop = P().PseudoQuantOp(192, 8, 16, 0.9, 100, ntop, 0.1, gctx.addr(), 0, False)
    cop = O.PseudoQuantOp(192, 8, 16, 0.9, 100, ntop, 0.1, octx.addr())
    xg = x.to(DEV)
    og = gop.forward(xg, weight.to(DEV), count.to(DEV), False)
    oc = cop.forward(x, weight, count, False)
    grads = [torch.randn(x.shape, generator=g) for _ in range(ntop)]
    gg = gop.backward([t.to(DEV) for t in grads], xg, og[0])
    gc = cop.backward(grads, x, oc[0])
    close(gg[0], gc[0], 1e-6)
    close(gg[1], gc[1], 1e-4)
    assert torch.equal(gg[2].cpu(), gc[2])
    widths = O.widths_v3(W16, 16, 32, 64)
    for t in range(16):
        assert gg[0][t::16, :, :, int(widths[t]):].abs().max().item() == 0 if widths[t] < 64 else True
    if ntop == 1:   # straight-through: the value gradient passes unchanged inside the valid columns
        for t in range(16):
            v = int(widths[t])
            assert torch.equal(gg[0][t::16, :, :, :v].cpu(), grads[0][t::16, :, :, :v])


@pytest.mark.parametrize("near", [False, True])
def test_projects_backward(near):
    th = [-0.5, 0, 0.5, 1, -0.5, 0, 0.5, 1, -0.5, 0, 0.5, 1, 0, 0]
    ph = [0, 0, 0, 0, 0.25, 0.25, 0.25, 0.25, -0.25, -0.25, -0.25, -0.25, 0.5, -0.5]
    g = torch.Generator().manual_seed(22)
    x = torch.randn(1, 3, 256, 512, generator=g)
    gop, cop = P().ProjectsOp(171, 256, th, ph, 0.5, near, 0, False), O.ProjectsOp(171, 256, th, ph, 0.5, near)
    yg, yc = gop.forward(x.to(DEV))[0], cop.forward(x)[0]
    grad = torch.randn(yc.shape, generator=g)
    (gg, cg), (gc, cc) = gop.backward(grad.to(DEV)), cop.backward(grad)
    close(gg, gc, 1e-4)      # float atomics: the order of the sums is not defined
    close(cg, cc, 1e-4)
