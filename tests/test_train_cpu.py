"""Training path on the CPU: the operator layer runs on the oracle backend
(test infrastructure), the model / loop / sampler code is the product's."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_balanced_sampler_meets_the_mean_in_every_window():
    from pseudocylindrical_convolution_amd.SphereDataset import (MyDistributeSampler, SyntheticSphereDataSet,
                                                                 balance_windows)
    data = SyntheticSphereDataSet(96, 16, 32, seed=3)
    values = data.values()
    mean = float(np.mean(list(values.values())))
    world, batch, acc = 2, 2, 3
    ws = world * batch * acc
    seen = []
    for rank in range(world):
        s = MyDistributeSampler(data, world, rank, batch, True, 5, mean=0.93 * mean, acc_batch=acc, values=values)
        s.set_epoch(2)
        order = s.global_order()
        mine = list(iter(s))
        assert mine == order[rank::world]
        seen.append(mine)
        assert sorted(order) == list(range(96))                      # still a permutation
        for w in range(len(order) // ws):
            tot = sum(values[data.img_list[i]] for i in order[w * ws:(w + 1) * ws])
            assert tot >= 0.93 * mean * ws
    assert not set(seen[0]) & set(seen[1])
    # the plain shuffled order did need fixing, and another epoch gives another order
    g = torch.Generator().manual_seed(5 + 2)
    plain = torch.randperm(96, generator=g).tolist()
    sums = [sum(values[data.img_list[i]] for i in plain[w * ws:(w + 1) * ws]) for w in range(96 // ws)]
    assert min(sums) < 0.93 * mean * ws
    s.set_epoch(3)
    assert s.global_order() != order
    # an unreachable mean is reported, not looped on
    idx = list(range(12))
    assert balance_windows(idx, lambda i: 1.0, 4, 4.5) is False
    with pytest.raises(RuntimeError):
        bad = MyDistributeSampler(data, 1, 0, 4, True, 0, mean=10.0, values=values)
        bad.global_order()
    # no values: the plain DistributedSampler order
    s0 = MyDistributeSampler(data, 2, 0, 2, True, 5, values=None)
    s0.set_epoch(2)
    assert list(iter(s0)) == plain[0::2]


def test_rd_anchor_tables():
    from pseudocylindrical_convolution_amd.RDMetric import mse_tb, ssim_tb
    assert abs(float(mse_tb(0.167)) - 110.9652 / 255 / 255) < 1e-12 and abs(float(ssim_tb(2.3)) - 0.982) < 1e-12
    r = np.linspace(0.17, 2.2, 50)
    assert (np.diff(mse_tb(r)) < 0).all() and (np.diff(ssim_tb(r)) > 0).all()


def _tiny(oracle_backend, cls="CMPNetV2MF", **kw):
    from pseudocylindrical_convolution_amd import model_zoo_v2 as Z
    torch.manual_seed(0)
    return getattr(Z, cls)(8, 16, 16, 16, 8, False, kw.get("init", False), 0)


def test_entropy_net_is_causal_in_coding_order(oracle_backend):
    """the rate of a symbol may depend only on symbols on earlier wavefront planes (group + row +
    column, the order the codec's engine visits them): change one symbol and every rate that moves
    lies on a later plane -- or is the symbol's own"""
    from pseudocylindrical_convolution_amd import model_zoo_v2 as Z
    torch.manual_seed(1)
    net = Z.CMPNetV2MFEntropy(16, 16, 16, 16, 8, True, False, 0)       # 4 groups, optimised split
    g = torch.Generator().manual_seed(2)
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.06)
        net.ent.delta_net.net[6].conv.bias.fill_(2)
    h, w = 2, 64
    sym = torch.randint(0, 8, (16, 4, h, w), generator=g).float()
    with torch.no_grad():
        base, mask = net(sym.clone())
        base, mask = base.view(16, 4, h, w).clone(), mask.view(16, 4, h, w).clone()
    assert mask.sum() > 0 and torch.isfinite(base).all() and (base * mask >= -1e-5).all()
    plane = (torch.arange(16).view(16, 1, 1, 1) * h + torch.arange(h).view(1, 1, h, 1) +
             torch.arange(w).view(1, 1, 1, w) + torch.arange(4).view(1, 4, 1, 1)).expand(16, 4, h, w)
    for (t, c, i, j) in [(3, 1, 0, 5), (8, 0, 1, 20), (0, 3, 1, 2), (15, 2, 0, 7), (7, 3, 1, 40)]:
        assert mask[t, c, i, j] == 1
        other = sym.clone()
        other[t, c, i, j] = (other[t, c, i, j] + 3) % 8
        with torch.no_grad():
            moved = (net(other)[0].view(16, 4, h, w) != base)
        assert moved[t, c, i, j]                                       # its own rate: the label changed
        moved[t, c, i, j] = False
        assert moved.any()
        assert (plane[moved] > plane[t, c, i, j]).all()


def test_training_step_reaches_every_parameter_and_lowers_the_loss(oracle_backend):
    net = _tiny(oracle_backend)
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=2e-3)
    x = torch.rand(1, 3, 256, 512, generator=torch.Generator().manual_seed(3))
    losses = []
    for it in range(6):
        y, ent, mask = net(x)
        assert y.shape == x.shape and ent.shape == mask.shape
        loss = torch.mean((y - x) ** 2) + 0.01 * torch.sum(ent) / torch.sum(mask).item()
        opt.zero_grad()
        loss.backward()
        if it == 0:
            missing = [n for n, p in net.named_parameters() if p.grad is None]
            assert not missing, missing
            hist = net.quant.count.grad
            # the "gradient" of quant.count is minus the per-channel histogram of this call: SGD adds it
            assert hist.max() <= 0 and (hist.sum(1) == hist.sum(1)[0]).all() and hist.sum(1)[0] < 0
            assert all(torch.isfinite(p.grad).all() for p in net.parameters())
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0]
    w = net.ent.weight_net.net[0].conv.weight
    net.ent.weight_net.net[0].conv(torch.zeros(16, 2, 6, 68))           # forward re-applies the mask
    ref = w.detach().clone()
    oracle_backend.MaskConstrainOp(5, 2).forward(ref)
    assert torch.equal(ref, w.detach())


def test_init_stage_cuts_the_rate_gradient_off_the_codes(oracle_backend):
    net = _tiny(oracle_backend, init=True)
    x = torch.rand(1, 3, 256, 512, generator=torch.Generator().manual_seed(4))
    _, ent, mask = net(x)
    (torch.sum(ent) / torch.sum(mask).item()).backward()
    enc_grads = [p.grad for p in net.encoder.parameters() if p.grad is not None]
    assert all(g.abs().max().item() == 0 for g in enc_grads)
    assert any(p.grad is not None and p.grad.abs().max().item() > 0 for p in net.ent.parameters())


def _rank(rank, world, port, base_dir, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from pseudocylindrical_convolution_amd.PCONV_operator import backend
    from oracle import pconv_cpu, coder_cpu
    backend.use(pconv_cpu, coder_cpu)
    pconv_cpu.set_detmath(True)
    torch.set_num_threads(2)
    from pseudocylindrical_convolution_amd import train
    args = train.build_parser().parse_args(
        ["--device", "cpu", "--synthetic", "8", "--height", "256", "--width", "512", "--batch-size", "1",
         "--test-batch-size", "1", "--acc-batch", "2", "--epochs", "2", "--valid-dim", "8", "--channels", "16",
         "--code-dim", "16", "--viewport_size", "24", "--lr", "0.001", "--mean", "1.0", "--workers", "0",
         "--no-opt", "--clip", "1.0", "--base-dir", base_dir, "--max-steps", "4"])
    hist = train.Job(rank, world, args)
    ret[rank] = hist


@pytest.mark.timeout(900)
def test_two_rank_ddp_training_job(tmp_path):
    """the reference's Job on two gloo ranks: an epoch on the transforms, an epoch on the entropy
    model, evaluation, checkpointing; ranks stay in step"""
    world, port = 2, 29000 + os.getpid() % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rank, args=(world, port, str(tmp_path), ret), nprocs=world, join=True)
    h0, h1 = ret[0], ret[1]
    assert len(h0) == 2
    for (last, ls) in h0:
        assert all(np.isfinite(v) for v in last) and np.isfinite(ls[0])
    assert [ls for _, ls in h0] == [ls for _, ls in h1]                # same model on both ranks
    saved = sorted(os.listdir(os.path.join(str(tmp_path), "save_models")))
    assert "ent_normal_16_8_16_best_0.pt" in saved and any(s.endswith("_logs_0.txt") for s in saved)
    sd = torch.load(os.path.join(str(tmp_path), "save_models", "ent_normal_16_8_16_best_0.pt"))
    assert any(k.startswith("ent.weight_net.net.0.conv.weight") for k in sd) and "quant.count" in sd
    log = open(os.path.join(str(tmp_path), "save_models", "ent_normal_16_8_16_logs_0.txt")).read()
    assert "Train Epoch: 1" in log and "Train Epoch: 2" in log and "Test set:" in log


@pytest.mark.timeout(600)
def test_exported_checkpoint_codes_at_the_rate_the_training_graph_predicts(oracle_backend, tmp_path):
    """train-time graph -> export -> the codec: the three files load with strict=True, the code file
    is as long as the whole-tensor entropy model says (sum of -log2 p, + the coder's closing bytes),
    and decoding reproduces the training graph's reconstruction bit for bit.  Two independent
    restatements meet here: masked whole-tensor convolutions with the causal pad, and the wavefront
    engine ops with their causal halo lists."""
    import math
    from pseudocylindrical_convolution_amd import export, model_zoo_v2 as Z, pseudo_codec as PC
    torch.manual_seed(0)
    vd = 8
    net = Z.CMPNetV2MF(vd, 192, 192, 16, 8, True, False, 0)
    g = torch.Generator().manual_seed(2)
    with torch.no_grad():
        for p in net.ent.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.1)
        net.ent.delta_net.net[6].conv.bias.fill_(1.0)
    net.eval()
    x = torch.rand(1, 3, 512, 1024, generator=g)
    with torch.no_grad():
        y, ent, mask = net(x)
    bits = ent.sum().item() / math.log(2)
    paths = export.export_codec(net, vd, str(tmp_path), "t")
    enc, dec = PC.PseudoEncoder(vd, 0), PC.PseudoDecoder(vd, 0)
    PC.load_models(enc, paths[0], paths[2], "cpu")                       # strict
    PC.load_models(dec, paths[1], paths[2], "cpu")
    code = str(tmp_path / "code.bin")
    enc(x, code)
    coded = os.path.getsize(code) * 8
    assert mask.sum().item() == 13376
    assert -8 <= coded - bits <= 0.002 * bits + 40, (coded, bits)
    assert torch.equal(dec(code, 512, 1024), y)
    # a DDP-style state dict ("module." prefix) exports the same tensors
    wrapped = {"module." + k: v for k, v in net.state_dict().items()}
    again = export.export_codec(wrapped, vd, str(tmp_path / "b"), "t")
    a, b = torch.load(paths[2]), torch.load(again[2])
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)
    assert a["ent.net.0.conv.weight"].shape == (3, 6, 2, 5, 5)


def _init_stage_job(base_dir, extra, port):
    import torch.distributed as dist
    from pseudocylindrical_convolution_amd import train
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    args = train.build_parser().parse_args(
        ["--device", "cpu", "--synthetic", "2", "--height", "256", "--width", "512", "--batch-size", "1",
         "--test-batch-size", "1", "--epochs", "0", "--valid-dim", "8", "--channels", "16", "--code-dim", "16",
         "--viewport_size", "24", "--workers", "0", "--no-opt", "--base-dir", base_dir, "--init"] + extra)
    try:
        return train.Job(0, 1, args)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_init_stage_starts_from_the_base_stage_checkpoint(oracle_backend, tmp_path):
    """--init (entropy model on frozen transforms) loads what --base wrote
    (trainDDP_Full.py:118-122: save_models/base_opt_192_{valid_dim}_16_best_0.pt) and refuses to go on
    from random transforms when nothing is there"""
    port = 31000 + os.getpid() % 2000
    with pytest.raises(FileNotFoundError):
        _init_stage_job(str(tmp_path), [], port)
    assert _init_stage_job(str(tmp_path), ["--init-random"], port + 1) == []
    base = _tiny(oracle_backend, cls="CMPNetV2M")
    os.makedirs(os.path.join(str(tmp_path), "save_models"), exist_ok=True)
    torch.save(base.state_dict(), os.path.join(str(tmp_path), "save_models", "base_normal_16_8_16_best_0.pt"))
    assert _init_stage_job(str(tmp_path), [], port + 2) == []
    log = open(os.path.join(str(tmp_path), "save_models", "ent_normal_16_8_16_init_logs_0.txt")).read()
    assert "base_normal_16_8_16_best_0.pt successful" in log


def test_procedural_images_and_packed_weights():
    """(r6) the procedural training images are deterministic functions of (seed, index) in [0, 1] with real
    structure (flat shapes: many repeated values; texture: not constant), and the packed checkpoint format keeps
    small tensors bit for bit and the large ones to fp16 precision"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import weights_pack
    from pseudocylindrical_convolution_amd.SphereDataset import ProceduralSphereDataSet
    a, b = ProceduralSphereDataSet(4, 64, 128, seed=3), ProceduralSphereDataSet(4, 64, 128, seed=3)
    x = a[2]
    assert x.shape == (3, 64, 128) and x.dtype == torch.float32 and 0.0 <= x.min() and x.max() <= 1.0
    assert torch.equal(x, b[2]) and not torch.equal(x, a[1]) and not torch.equal(x, ProceduralSphereDataSet(4, 64, 128, seed=4)[2])
    assert x.std() > 0.02 and set(a.values()) == set(a.img_list) and all(0.5 <= v <= 2.5 for v in a.values().values())
    state = {"big": torch.randn(300, 300), "small": torch.randn(7), "count": torch.arange(5)}
    path = os.path.join(ROOT, "gpurun_out", "_pack_test.pt")
    try:
        weights_pack.pack(path, {"s": state})
        back = weights_pack.unpack(path)["s"]
    finally:
        if os.path.exists(path):
            os.remove(path)
    assert back["big"].dtype == torch.float32 and torch.equal(back["big"], state["big"].half().float())
    assert torch.equal(back["small"], state["small"]) and torch.equal(back["count"], state["count"])


def test_time_budget_ends_a_base_stage_run_and_keeps_its_checkpoint(oracle_backend, tmp_path):
    """(r6) --procedural N --time-budget S: a --base run with far more epochs than fit stops at the budget, having
    tested and saved what it has (the round's real training runs inside 20-minute GPU calls this way)"""
    import time
    import torch.distributed as dist
    from pseudocylindrical_convolution_amd import train
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(33000 + os.getpid() % 2000), RANK="0", WORLD_SIZE="1")
    args = train.build_parser().parse_args(
        ["--device", "cpu", "--procedural", "4", "--height", "256", "--width", "512", "--batch-size", "1",
         "--test-batch-size", "1", "--acc-batch", "1", "--epochs", "100000", "--time-budget", "4", "--valid-dim", "8",
         "--channels", "16", "--code-dim", "16", "--viewport_size", "24", "--workers", "0", "--no-opt", "--mean", "0",
         "--base-dir", str(tmp_path), "--base"])
    t0 = time.time()
    try:
        hist = train.Job(0, 1, args)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
    assert 1 <= len(hist) < 1000 and time.time() - t0 < 120
    assert os.path.exists(os.path.join(str(tmp_path), "save_models", "base_normal_16_8_16_best_0.pt"))
    log = open(os.path.join(str(tmp_path), "save_models", "base_normal_16_8_16_logs_0.txt")).read()
    assert "time budget" in log
