"""Training path on the MI355X: the whole-tensor entropy model and the end-to-end
graph through the HIP ops' forward + backward, against the same modules on the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import pconv_cpu as O
from oracle import coder_cpu
from pseudocylindrical_convolution_amd.PCONV_operator import backend

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _entropy_net(dev, seed=1, init=False):
    from pseudocylindrical_convolution_amd import model_zoo_v2 as Z
    torch.manual_seed(seed)
    net = Z.CMPNetV2MFEntropy(56, 192, 192, 16, 8, True, init, 0)         # 14 groups: the benchmark's model
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.04)
        net.ent.delta_net.net[6].conv.bias.fill_(2)
    return net.to(dev)


def test_entropy_net_rate_and_gradients_match_the_oracle():
    """rate per symbol within 2e-3 (the -log of an fp32 probability), parameter gradients within
    1e-2 of each tensor's largest gradient (sums of 1/p-amplified fp32 terms over every symbol, in a
    different order on each side); the conv itself is the vendor library's on both sides"""
    O.set_detmath(True)
    h, w = 4, 128
    sym = torch.randint(0, 8, (16, 14, h, w), generator=torch.Generator().manual_seed(5)).float()

    def run(dev):
        net = _entropy_net(dev)
        ent, mask = net(sym.clone().to(dev))
        (torch.sum(ent) / torch.sum(mask).item()).backward()
        grads = {n: p.grad.detach().cpu().clone() for n, p in net.named_parameters()}
        return ent.detach().cpu(), mask.detach().cpu(), grads

    backend.reset()
    eg, mg, gg = run(DEV)
    backend.use(O, coder_cpu)
    try:
        ec, mc, gc = run("cpu")
    finally:
        backend.reset()
    assert torch.equal(mg, mc) and mg.sum() > 0
    assert (eg - ec).abs().max().item() <= 2e-3 * max(1.0, ec.abs().max().item())
    assert abs(eg.sum().item() - ec.sum().item()) <= 1e-4 * ec.sum().item()
    assert set(gg) == set(gc)
    for name in gc:
        scale = max(gc[name].abs().max().item(), 1e-6)
        assert (gg[name] - gc[name]).abs().max().item() <= 1e-2 * scale + 1e-7, name


def test_end_to_end_training_steps_on_the_gpu():
    """CMPNetV2MF at the benchmark's width (192 channels, valid_dim 56) on 512x1024: every parameter
    gets a finite gradient through the HIP backward ops and a few Adam steps lower the loss"""
    from pseudocylindrical_convolution_amd import model_zoo_v2 as Z
    from pseudocylindrical_convolution_amd.PCONV_operator import MultiProject
    backend.reset()
    torch.manual_seed(0)
    net = Z.CMPNetV2MF(56, 192, 192, 16, 8, True, False, 0).to(DEV)
    net.train()
    # two projectors, as in the reference's loop: an op object owns (and re-returns) its output buffer
    pr1, pr2 = MultiProject(96, 144, 0.5, False, 0).to(DEV), MultiProject(96, 144, 0.5, False, 0).to(DEV)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    x = torch.rand(2, 3, 512, 1024, generator=torch.Generator().manual_seed(3)).to(DEV)
    losses = []
    for it in range(5):
        y, ent, mask = net(x)
        py, px = pr1(y), pr2(x)
        loss = torch.mean((px - py) ** 2) + 0.05 * torch.sum(ent) / torch.sum(mask).item()
        opt.zero_grad()
        loss.backward()
        if it == 0:
            missing = [n for n, p in net.named_parameters() if p.grad is None]
            assert not missing, missing
            assert all(torch.isfinite(p.grad).all().item() for p in net.parameters())
            assert net.encoder.net[0].conv1.weight.grad.abs().max().item() > 0
        opt.step()
        losses.append(loss.item())
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
    # the same weights drive the codec's analysis transform: eval-mode forward runs the MFMA path
    net.eval()
    with torch.no_grad():
        code = net.encoder(net.slice(x[:1]))
        assert code.shape == (16, 192, 2, 64) and torch.isfinite(code).all()


def test_exported_checkpoint_codes_at_the_predicted_rate_on_the_engine(tmp_path):
    """the benchmark's model (valid_dim 56): rate predicted by the whole-tensor EntropyNet on the
    GPU vs the bytes the native entropy engine writes, and decode == the training graph's output"""
    import math
    from pseudocylindrical_convolution_amd import export, model_zoo_v2 as Z, pseudo_codec as PC
    backend.reset()
    torch.manual_seed(0)
    vd = 56
    net = Z.CMPNetV2MF(vd, 192, 192, 16, 8, True, False, 0).to(DEV)
    g = torch.Generator().manual_seed(2)
    with torch.no_grad():
        for p in net.ent.parameters():
            p.copy_((torch.randn(p.shape, generator=g) * 0.05).to(DEV))
        net.ent.delta_net.net[6].conv.bias.fill_(1.0)
    net.eval()
    x = torch.rand(1, 3, 512, 1024, generator=g).to(DEV)
    with torch.no_grad():
        y, ent, mask = net(x)
    bits = ent.sum().item() / math.log(2)
    paths = export.export_codec(net, vd, str(tmp_path), "t")
    enc, dec = PC.PseudoEncoder(vd, 0).to(DEV), PC.PseudoDecoder(vd, 0).to(DEV)
    PC.load_models(enc, paths[0], paths[2], DEV)
    PC.load_models(dec, paths[1], paths[2], DEV)
    code = str(tmp_path / "code.bin")
    enc(x, code)
    coded = os.path.getsize(code) * 8
    assert mask.sum().item() == 13376 * 7
    assert -8 <= coded - bits <= 0.002 * bits + 40, (coded, bits)
    out = dec(code, 512, 1024)
    assert (out - y).abs().max().item() <= 1e-4
