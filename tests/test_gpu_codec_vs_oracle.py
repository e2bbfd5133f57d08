"""GPU: the whole codec and the remaining op entry points against the CPU oracle.

* native engine (bulk encoder, step decoder) vs the oracle run of the same graph at
  the reference-native 512x1024 (12 layers, latent 4 x 128), at the metric size
  2048x4096 (BASELINE config #4 as written) and for a two-frame lock-step batch:
  symbols and bitstream bit for bit, reconstruction <= 1e-4;
* BASELINE config #2 as written: SphereSlice -> PseudoPadV2(p) -> SphereUslice(pad=p)
  at 1x3x512x1024;
* the non-batch entry points of EntropyGmmTableOp / EntropyConv2Op (a22);
* PseudoQuantOp.forward(train=True) across a check_iters boundary, and its
  per-call histogram;
* viewport PSNR / SSIM at the metric size (2048x4096).
The oracle runs take tens of seconds each on the GPU box's host cores."""
import os

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


def P():
    from pseudocylindrical_convolution_amd import PCONV
    return PCONV


def same(a_gpu, b_cpu):
    a = a_gpu.detach().cpu()
    assert a.shape == b_cpu.shape
    assert torch.equal(a, b_cpu), "max abs diff %g" % (a - b_cpu).abs().max().item()


def _codec(device_id=0):
    from pseudocylindrical_convolution_amd import pseudo_codec as PC
    torch.manual_seed(1234)
    enc, dec = PC.PseudoEncoder(56, device_id).eval(), PC.PseudoDecoder(56, device_id).eval()
    g = torch.Generator().manual_seed(7)
    sd = {k: torch.randn(v.shape, generator=g) * 0.05 for k, v in enc.ent.state_dict().items()}
    enc.ent.load_state_dict(sd)
    dec.ent.load_state_dict(sd)
    dec.quant.weight.data.copy_(enc.quant.weight.data)
    return enc, dec


def _oracle_codec(frames, H, W, tmp_path):
    """frames (n,3,H,W) on the CPU -> per frame (symbols, stream bytes, reconstruction)"""
    from pseudocylindrical_convolution_amd.PCONV_operator import backend
    from oracle import coder_cpu
    backend.use(O, coder_cpu)
    O.set_detmath(True)
    out = []
    try:
        enc, dec = _codec()
        for i in range(frames.shape[0]):
            x = frames[i:i + 1].contiguous()
            path = str(tmp_path / ("cpu%d.bin" % i))
            enc(x, path)
            sym = enc.ent.fill(enc.symbols(x)).clone()
            rec = dec(path, H, W).clone()
            with open(path, "rb") as f:
                out.append((sym, f.read(), rec))
    finally:
        backend.reset()
    return out


@pytest.mark.timeout(1500)
def test_engine_equals_oracle_at_reference_size(hip_backend, tmp_path):
    """512x1024 (pseudo_codec.py:229-234): the 12-layer model on a 4 x 128 symbol plane per
    tile, 204 steps.  Engine encode -> oracle's bytes; engine decode of the ORACLE's bytes ->
    oracle's symbols and image."""
    from pseudocylindrical_convolution_amd.engine import CodecEngine
    H, W = 512, 1024
    x = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(3))
    (csym, cbytes, crec), = _oracle_codec(x, H, W, tmp_path)
    enc, dec = _codec()
    eng = CodecEngine(56, 0, enc, dec)
    gsym = eng.symbols(x.cuda())
    assert tuple(gsym.shape) == (16, 14, 4, 128)
    # the analysis transform is fp32 work (Winograd by default): a quantiser input on a decision level may
    # round to the other symbol.  Counted (0 observed), never assumed; the entropy stage below is exact.
    ties = int((gsym.cpu() != csym).sum())
    assert ties <= 4, "%d of %d symbols differ from the oracle's" % (ties, csym.numel())
    e = eng._engine("enc", 4, 128, 1)
    streams = e.encode(csym.cuda().contiguous())           # the ORACLE's symbols -> the oracle's bytes, always
    assert streams[0] == cbytes, "engine stream differs from the oracle's (%d vs %d bytes)" % (len(streams[0]), len(cbytes))
    if ties == 0:
        assert eng.encode(x.cuda())[0] == cbytes
    out = eng._engine("dec", 4, 128, 1).decode([cbytes])
    same(out, csym)
    rec = eng.decode([cbytes], H, W).cpu()
    err = (rec - crec).abs().max().item()
    assert err < 1e-4, "reconstruction differs from the oracle by %g" % err


def _metric_frame(seed=3):
    """smooth content + noise, so that the rate is not the worst case and all 8 levels occur"""
    H, W = 2048, 4096
    x = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(seed))
    yy = torch.linspace(0, 1, H).view(1, 1, H, 1)
    xx = torch.linspace(0, 1, W).view(1, 1, 1, W)
    k = 3 + (seed - 3) % 5
    return (0.5 + 0.3 * torch.sin(6.28318 * k * xx + 0.37 * (seed - 3)) * torch.cos(3.14159 * 2 * yy)
            + 0.1 * (x - 0.5)).clamp_(0, 1).contiguous()


@pytest.fixture(scope="module")
def metric_oracle(tmp_path_factory):
    """ONE oracle run of BASELINE config #4's frame (1x3x2048x4096; ~70 s on the GPU box's host cores),
    shared by the tests of this module: analysis codes, symbols, stream, decoded symbols, reconstruction"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pseudocylindrical_convolution_amd.PCONV_operator import backend
    from oracle import coder_cpu
    H, W = 2048, 4096
    x = _metric_frame()
    path = str(tmp_path_factory.mktemp("metric") / "cpu.bin")
    backend.use(O, coder_cpu)
    O.set_detmath(True)
    torch_threads = torch.get_num_threads()
    torch.set_num_threads(O.set_num_threads())                     # the box's CPU share, not its 256 visible CPUs
    try:
        cenc, cdec = _codec()
        with torch.no_grad():
            ccode = cenc.encoder(cenc.slice(x)).clone()
            _, code_i = cenc.quant(ccode)
            csym = cenc.ent.fill(cenc.dtw(cenc.ext(code_i))).clone()
            live = int(cenc.ent.fill(torch.ones_like(csym)).sum())
        cenc.ent.start(path)
        cenc.ent(csym)
        with open(path, "rb") as f:
            cbytes = f.read()
        cdec.ent.start(path)
        cback = cdec.ent(2 * (H // 256), 2 * (W // 16)).clone()
        crec = cdec.reconstruct(cback).clone()
    finally:
        backend.reset()
        torch.set_num_threads(torch_threads)
    assert torch.equal(cback, csym)                                # the oracle round-trips its own stream
    assert tuple(csym.shape) == (16, 14, 16, 512)
    assert len(csym.unique()) >= 3
    return {"x": x, "code": ccode, "sym": csym, "bytes": cbytes, "rec": crec, "live": live}


@pytest.mark.timeout(1700)
def test_engine_equals_oracle_at_the_metric_size(hip_backend, metric_oracle):
    """BASELINE config #4 as written: 1x3x2048x4096 (W x H = 4096 x 2048), model-idx 3 --ssim.
    780 wavefront steps, 1.5 M symbols, the regime where the reference's fp32-stored offsets and
    32-bit byte offsets break.  Against the module's one oracle run:
      * analysis codes <= 1e-4, quantiser symbols equal up to float ties at a decision level
        (counted; the entropy comparison below does not depend on them),
      * the engine codes the ORACLE's symbols into the ORACLE's bytes,
      * the engine decodes the oracle's bytes into the oracle's symbols,
      * HIP synthesis of those symbols <= 1e-4 from the oracle's reconstruction."""
    from pseudocylindrical_convolution_amd.engine import CodecEngine
    H, W = 2048, 4096
    x, ccode, csym, cbytes, crec, live = (metric_oracle[k] for k in ("x", "code", "sym", "bytes", "rec", "live"))
    enc, dec = _codec()
    eng = CodecEngine(56, 0, enc, dec)
    with torch.no_grad():
        gcode = enc.encoder(enc.slice(x.cuda())).cpu()
    code_err = (gcode - ccode).abs().max().item()
    assert code_err <= 1e-4
    gsym = eng.symbols(x.cuda()).cpu()
    ties = int((gsym != csym).sum())
    assert ties <= 16, "%d of %d symbols differ from the oracle's" % (ties, csym.numel())

    e = eng._engine("enc", 16, 512, 1)
    assert e.symbols_per_image == live
    streams = e.encode(csym.cuda().contiguous())
    assert len(streams[0]) == len(cbytes), "engine stream: %d bytes, oracle: %d" % (len(streams[0]), len(cbytes))
    assert streams[0] == cbytes
    if ties == 0:
        assert eng.encode(x.cuda())[0] == cbytes
    out = eng._engine("dec", 16, 512, 1).decode([cbytes])
    same(out, csym)
    rec = eng.decode([cbytes], H, W).cpu()
    err = (rec - crec).abs().max().item()
    # the ACTUAL figures behind the 1e-4 assertions, for profiles/: default kernels (Winograd
    # F(4x2) / F(2x2) + the direct 1x1 / stride-2 kernels) against the oracle's fmaf-chain convolution
    _record("metric_size_2048x4096", {"analysis_code_max_abs_err": code_err, "code_scale_max_abs": ccode.abs().max().item(),
                                      "quantiser_ties": ties, "symbols": csym.numel(), "reconstruction_max_abs_err": err,
                                      "reconstruction_rms_err": (rec - crec).pow(2).mean().sqrt().item(),
                                      "conv3x3": os.environ.get("PCONV_CONV3X3", "wino42 (default)")})
    assert err < 1e-4, "reconstruction differs from the oracle by %g" % err


def _record(key, value):
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "gpurun_out", "codec_vs_oracle_errors.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[key] = value
        with open(path, "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)
    except (OSError, ValueError):
        pass


@pytest.mark.timeout(1700)
def test_benchmarked_workload_eight_frames_at_the_metric_size(hip_backend, metric_oracle, monkeypatch):
    """The workload bench.py times (BASELINE config #5's per-GPU share): EIGHT 2048x4096 frames through
    CodecEngine with its default knobs -- analysis blocks [0, 3) frame by frame, the rest batched over
    the call (tile batch 128, 1/4-scale tensors of 3.4 GB, the regime of pseudo_pad.cu:106's int32
    count); entropy encode in chunks of two; decode as four host-driven lock-step groups of two;
    synthesis blocks [0, 8) batched.  Everything the batch produces equals, bit for bit, what the same
    engine produces frame by frame, and frame 0 -- the oracle's frame -- meets the oracle's run."""
    from pseudocylindrical_convolution_amd.engine import CodecEngine
    for name in ("PCONV_ANALYSIS_SPLIT", "PCONV_SYNTHESIS_SPLIT", "PCONV_ENGINE_GROUPS", "PCONV_ENGINE_CHAIN",
                 "PCONV_ENCODE_CHUNK", "PCONV_DECODE_CHUNK", "PCONV_CONV3X3"):
        monkeypatch.delenv(name, raising=False)
    H, W, F = 2048, 4096, 8
    enc, dec = _codec()
    eng = CodecEngine(56, 0, enc, dec)
    assert (eng.ANALYSIS_SPLIT, eng.SYNTHESIS_SPLIT, eng.ENCODE_CHUNK, eng.DECODE_CHUNK) == (3, 8, 2, 0)
    frames = torch.cat([metric_oracle["x"]] + [_metric_frame(3 + i) for i in range(1, F)], 0).cuda()

    sym_b = eng.symbols(frames)                      # batched tail
    assert tuple(sym_b.shape) == (16 * F, 14, 16, 512)
    streams_b = eng.encode(frames)                   # chunks of two through the bulk encoder
    dec_b = eng._engine("dec", 16, 512, F).decode(streams_b)   # four host-driven groups of two
    assert torch.equal(dec_b, sym_b), "decoded symbols of the batch differ from the encoded ones"
    rec_b = eng.decode(streams_b, H, W)              # the same decode + batched synthesis head
    assert tuple(rec_b.shape) == (F, 3, H, W)

    for i in range(F):
        xi = frames[i:i + 1]
        sym_i = eng.symbols(xi)                      # whole transform on one frame
        assert torch.equal(sym_b[16 * i:16 * (i + 1)], sym_i), "frame %d: batched symbols differ" % i
        stream_i = eng.encode(xi)[0]
        assert stream_i == streams_b[i], "frame %d: batched stream differs (%d vs %d bytes)" % (
            i, len(streams_b[i]), len(stream_i))
        if i in (0, 3, 7):                           # one-frame decodes are 0.1 s each; three are enough
            rec_i = eng.decode([stream_i], H, W)     # queued chain, one group, per-frame synthesis
            assert torch.equal(rec_b[i:i + 1], rec_i), "frame %d: batched reconstruction differs" % i

    # frame 0 against the oracle: symbols up to counted ties, and -- when there are none -- the same bytes;
    # the reconstruction within the north-star tolerance either way
    csym, cbytes, crec = metric_oracle["sym"], metric_oracle["bytes"], metric_oracle["rec"]
    ties = int((sym_b[:16].cpu() != csym).sum())
    assert ties <= 16, "%d symbols of frame 0 differ from the oracle's" % ties
    if ties == 0:
        assert streams_b[0] == cbytes
        assert (rec_b[:1].cpu() - crec).abs().max().item() < 1e-4
    # ... and unconditionally: the ORACLE's stream of frame 0 beside the batch's other seven through the same
    # eight-frame path (four host-driven groups, batched synthesis head) decodes to the oracle's symbols and
    # reconstructs within the tolerance, whatever the analysis side's ties were
    mixed = [cbytes] + list(streams_b[1:])
    dec_m = eng._engine("dec", 16, 512, F).decode(mixed)
    assert torch.equal(dec_m[:16].cpu(), csym), "the oracle's stream decodes differently inside the batch"
    assert torch.equal(dec_m[16:], sym_b[16:])
    rec_m = eng.decode(mixed, H, W)
    err = (rec_m[:1].cpu() - crec).abs().max().item()
    assert err < 1e-4, "batched reconstruction of the oracle's stream differs from the oracle by %g" % err
    assert torch.equal(rec_m[1:], rec_b[1:])
    # bit rates of the eight frames are sane and differ (different content)
    assert len(set(len(s) for s in streams_b)) > 1 and all(len(s) > 1000 for s in streams_b)


@pytest.mark.timeout(1500)
def test_engine_lockstep_pair_equals_oracle(hip_backend, tmp_path):
    """two frames in lock-step (two groups / one group of two) vs the oracle coding them one by one"""
    from pseudocylindrical_convolution_amd.engine import CodecEngine
    H, W = 256, 512
    x = torch.rand(2, 3, H, W, generator=torch.Generator().manual_seed(5))
    ref = _oracle_codec(x, H, W, tmp_path)
    enc, dec = _codec()
    eng = CodecEngine(56, 0, enc, dec)
    streams = eng.encode(x.cuda())
    assert streams == [r[1] for r in ref]
    sym = eng._engine("dec", 2, 64, 2).decode(streams)
    same(sym, torch.cat([r[0] for r in ref], 0))
    rec = eng.decode(streams, H, W).cpu()
    for i in range(2):
        assert (rec[i:i + 1] - ref[i][2]).abs().max().item() < 1e-4
    os.environ["PCONV_ENGINE_GROUPS"] = "1"   # both frames inside ONE lock-step group
    try:
        eng1 = CodecEngine(56, 0, enc, dec)
        assert eng1.encode(x.cuda()) == streams
        same(eng1._engine("dec", 2, 64, 2).decode(streams), torch.cat([r[0] for r in ref], 0))
    finally:
        del os.environ["PCONV_ENGINE_GROUPS"]


def test_engine_follows_reloaded_weights(hip_backend, tmp_path):
    """the engine repacks the weights it owns when the entropy model is reloaded or edited
    after the first encode (cached engines are re-bound on every fetch)"""
    from pseudocylindrical_convolution_amd.engine import CodecEngine
    from pseudocylindrical_convolution_amd.PCONV_operator import backend
    enc, dec = _codec()
    eng = CodecEngine(56, 0, enc, dec)
    x = torch.rand(1, 3, 256, 512, generator=torch.Generator().manual_seed(8)).cuda()
    first = eng.encode(x)[0]
    g = torch.Generator().manual_seed(99)
    sd = {k: torch.randn(v.shape, generator=g) * 0.05 for k, v in enc.ent.state_dict().items()}
    enc.ent.load_state_dict(sd)
    dec.ent.load_state_dict(sd)
    second = eng.encode(x)[0]
    assert second != first
    path = str(tmp_path / "perop.bin")
    enc.forward_per_op(x, path)
    with open(path, "rb") as f:
        assert second == f.read()
    sym = eng.symbols(x)
    same(eng._engine("dec", sym.shape[2], sym.shape[3], 1).decode([second]), sym.cpu())
    # a write through .data is invisible to _version: invalidate_derived() covers it
    enc.ent.net[0].conv.bias.data.add_(0.01)
    dec.ent.net[0].conv.bias.data.add_(0.01)
    backend.invalidate_derived()
    third = eng.encode(x)[0]
    enc.forward_per_op(x, path)
    with open(path, "rb") as f:
        assert third == f.read()
    same(eng._engine("dec", sym.shape[2], sym.shape[3], 1).decode([third]), sym.cpu())


@pytest.mark.parametrize("pad", [1, 2])
def test_slice_pad_uslice_chain_config2(pad):
    """BASELINE config #2: 1x3x512x1024, SphereSlice(pad 0) -> PseudoPadV2(p) -> SphereUslice(pad p)"""
    x = torch.rand(1, 3, 512, 1024, generator=torch.Generator().manual_seed(1))
    gctx = P().PseudoContextOp(16, 20, W16, 0, False)
    octx = O.PseudoContextOp(16, 20, W16)
    gs = P().SphereSliceOp(16, 0, 0, W16, 0, False).forward(x.to(DEV))[0]
    cs = O.SphereSliceOp(16, 0, 0, W16).forward(x)[0]
    same(gs, cs)
    gp = P().PseudoPadOp(pad, 16, gctx.addr(), 0, False).forward(gs)[0]
    cp = O.PseudoPadOp(pad, 16, octx.addr()).forward(cs)[0]
    assert tuple(gp.shape) == (16, 3, 32 + 2 * pad, 1024 + 2 * pad)
    same(gp, cp)
    gu = P().SphereUsliceOp(16, 0, pad, W16, 0, False).forward(gp)[0]
    cu = O.SphereUsliceOp(16, 0, pad, W16).forward(cp)[0]
    assert tuple(gu.shape) == (1, 3, 512, 1024)
    same(gu, cu)
    # the round trip is the identity (to rounding) where no resampling happens: the six
    # full-width tiles, rows 160..351
    assert (gu[:, :, 160:352].cpu() - x[:, :, 160:352]).abs().max().item() < 1e-5


def test_gmm_table_non_batch_entry():
    """EntropyGmmTableOp.forward(weight, delta, mean, tnum) (entropy_gmm_table_cuda.cu:59-80,107-133)"""
    n = 3000
    g = torch.Generator().manual_seed(41)
    wt = (torch.randn(n, 3, generator=g) * 2).contiguous()
    dl = (torch.randn(n, 3, generator=g).abs() * 1.5 - 0.1).contiguous()
    mu = (torch.randn(n, 3, generator=g) * 4).contiguous()
    tnum = torch.tensor([n - 7], dtype=torch.int32)
    wg, dg = wt.clone().to(DEV), dl.clone().to(DEV)
    tg = P().EntropyGmmTableOp(8, 3.5, 3, 65536, 1e-6, 0, False).forward(wg, dg, mu.to(DEV), tnum)[0]
    wc, dc = wt.clone(), dl.clone()
    tc = O.EntropyGmmTableOp(8, 3.5, 3, 65536, 1e-6).forward(wc, dc, mu, tnum)[0]
    same(tg[:n - 7], tc[:n - 7].contiguous())
    same(wg[:n - 7], wc[:n - 7].contiguous())   # softmax / delta written in place
    same(dg[:n - 7], dc[:n - 7].contiguous())
    rows = tg[:n - 7].cpu()
    assert (rows[:, 0] == 0).all() and (rows[:, 8] == 65536).all() and (rows[:, 1:] - rows[:, :-1] >= 1).all()


@pytest.mark.parametrize("act", [False, True])
def test_entropy_conv_non_batch_entries(act):
    """EntropyConv2Op.forward / forward_act (entropy_conv_cuda_v2.cu:61-235): one weight set,
    every wavefront step of a 2-frame batch"""
    ngroup, h, w, nimg = 4, 1, 64, 2

    def run(mod):
        dev = DEV if mod is not O else "cpu"
        ctx = mod.EntropyContextOp(16, 18, W16, 0, False)
        ctx.start_context(w)
        g = torch.Generator().manual_seed(51)
        wt = (torch.randn(ngroup * 3, ngroup, 5, 5, generator=g) * 0.2).to(dev)
        b = (torch.randn(ngroup * 3, generator=g) * 0.1).to(dev)
        a = torch.rand(ngroup * 3, generator=g).to(dev)
        data = torch.randint(0, 8, (nimg * 16, ngroup, h, w), generator=g).float().to(dev)
        data = mod.PseudoFillOp(0, 16, 0, 0, ctx.addr(), 2, 0, False).forward(data)[0]
        ipt = mod.DInput2Op(ngroup, 16, 2, -3.5, 1, ctx.addr(), 0, False)
        pad = mod.EntropyCtxPadRun2Op(2, 16, ngroup, True, ctx.addr(), 0, False)
        conv = mod.EntropyConv2Op(16, ngroup, ngroup, ngroup * 3, 5, 5, 2, 2, ctx.addr(), 0, False)
        lab = mod.DExtract2Op(16, ngroup, True, ctx.addr(), 0, False)
        label = torch.zeros((nimg, 1, h * 16, w), device=dev)
        y = None
        for _ in range(h * 16 + w + ngroup - 2):
            xin = pad.forward(ipt.forward(label)[0])[0]
            y = (conv.forward_act(xin, wt, b, a) if act else conv.forward(xin, wt, b))[0]
            label, _ = lab.forward(data)
        return y.cpu()

    yg, yc = run(P()), run(O)
    assert yg.abs().max().item() > 0
    assert torch.equal(yg, yc), "max abs diff %g" % (yg - yc).abs().max().item()


def test_quant_train_mode_update_weight_and_histogram():
    """PseudoQuantOp.forward(train=True): the level merge of update_weight fires at call
    check_iters (pseudo_quant_cuda.cu:97-143) and the op keeps the per-call histogram
    count_data_ (-1 per valid element, :64,83)"""
    check = 3
    g = torch.Generator().manual_seed(61)
    weight = torch.zeros(192, 8)
    weight[:, 0] = 1. / 9
    weight[:, 1:] = float(np.log(1. / 9))
    weight += torch.rand(192, 8, generator=g) * 0.05
    # a histogram that empties some top levels and, in a few channels, level 0
    count = torch.rand(192, 8, generator=g)
    count[::3, 5:] = 0
    count[::7, 0] = 0
    count[5::11, 1:] = 0
    gctx = P().PseudoContextOp(16, 20, W16, 0, False)
    octx = O.PseudoContextOp(16, 20, W16)
    gop = P().PseudoQuantOp(192, 8, 16, 0.9, check, 2, 0.1, gctx.addr(), 0, False)
    cop = O.PseudoQuantOp(192, 8, 16, 0.9, check, 2, 0.1, octx.addr())
    wg, cg = weight.clone().to(DEV), count.clone().to(DEV)
    wc, cc = weight.clone(), count.clone()
    for it in range(2 * check + 1):
        x = torch.rand(16, 192, 2, 64, generator=g) * 1.2 - 0.1
        gv, gi = gop.forward(x.to(DEV), wg, cg, True)
        cv, ci = cop.forward(x, wc, cc, True)
        if it > 0 and it % check == 0:
            # the merge ran inside this call, with torch's exp / log of either device: the
            # level tables agree to an ulp, so do the values; an index may flip only for an
            # input that sits on a decision boundary
            assert (gv.cpu() - cv).abs().max().item() < 1e-5
            assert (gi.cpu() != ci).float().mean().item() < 1e-4
        else:
            same(gi, ci)
            same(gv, cv)
            same(gop.count_data_, cop.count_data_)
        assert (wg.cpu() - wc).abs().max().item() < 1e-6, "level table diverged at call %d" % it
        assert (cg.cpu() - cc).abs().max().item() < 1e-6
        # the merge runs torch's exp / log on either device: keep the following kernel
        # comparisons exact by continuing from one copy of the (agreeing) tables
        wg.copy_(wc)
        cg.copy_(cc)
        # every valid element lands in exactly one bin
        assert int(-gop.count_data_.sum().item()) == 192 * 2 * 836
    assert not torch.equal(wc, weight) and not torch.equal(cc, count)   # the merge did fire


@pytest.mark.timeout(900)
def test_viewport_metrics_at_metric_size(hip_backend):
    """--test metric path (pseudo_codec.py:270-287) on one 2048x4096 frame: 14 viewports,
    PSNR and SSIM equal the oracle's ProjectsOp + torch SSIM on the same pair of images"""
    from pseudocylindrical_convolution_amd import pseudo_codec as PC
    from pseudocylindrical_convolution_amd.PCONV_operator import backend
    from oracle import coder_cpu
    H, W = 2048, 4096
    g = torch.Generator().manual_seed(71)
    yy = torch.linspace(0, 1, H).view(1, 1, H, 1)
    xx = torch.linspace(0, 1, W).view(1, 1, 1, W)
    a = (0.5 + 0.3 * torch.sin(12.566 * xx + 1.0) * torch.cos(6.283 * yy) + 0.1 * torch.rand(1, 3, H, W, generator=g))
    a = a.clamp_(0, 1).contiguous()
    b = (a + 0.02 * torch.randn(1, 3, H, W, generator=g)).clamp_(0, 1).contiguous()
    psnr_g, ssim_g = PC.ViewportMetrics(0)(a.cuda(), b.cuda())
    backend.use(O, coder_cpu)
    try:
        psnr_c, ssim_c = PC.ViewportMetrics(0)(a, b)
    finally:
        backend.reset()
    assert 25 < psnr_g < 45 and 0.5 < ssim_g < 1
    assert abs(psnr_g - psnr_c) < 1e-3 and abs(ssim_g - ssim_c) < 1e-4


@pytest.mark.timeout(1500)
def test_analysis_transform_config3_vs_oracle(hip_backend):
    """BASELINE config #3: SphereSlice + EncoderV2 (4 pseudo-conv stages + GDN) on 1x3x1024x2048,
    seeded default-init weights: the HIP path (fp32-MFMA tile convolutions, fused epilogues, ring
    pads) against the oracle-backend run of the same graph (torch CPU conv), <= 1e-4"""
    from pseudocylindrical_convolution_amd.PCONV_operator import backend
    from oracle import coder_cpu
    H, W = 1024, 2048
    x = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(2))
    enc, _ = _codec()
    with torch.no_grad():
        yg = enc.encoder(enc.slice(x.cuda())).cpu()
    backend.use(O, coder_cpu)
    try:
        cenc, _ = _codec()
        with torch.no_grad():
            yc = cenc.encoder(cenc.slice(x))
    finally:
        backend.reset()
    assert tuple(yg.shape) == (16, 192, 4, 128)
    err = (yg - yc).abs().max().item()
    assert err < 1e-4, "analysis transform differs from the oracle by %g" % err
    widths = O.widths_v3(W16, 16, 64, 128)
    for t in range(16):
        assert yg[t, :, :, int(widths[t]):].abs().max().item() == 0 if widths[t] < 128 else True
