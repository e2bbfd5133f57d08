"""GPU: the codec with TRAINED weights against the CPU oracle.

The model is the one tools/train_round6.py trains on the MI355X with the product's training path (train.py: stage 1
transforms + quantiser, stage 2 entropy model; procedural ERP images; export.py -> the codec's three files).  The
weights are not in the repository (a 35 MB blob): the packed file `trained/r6/codec_3_56.pack.pt` travels to the GPU
box with the tree when a training run of this round produced it, and these tests SKIP when it is absent (the driver's
round-end run on a fresh checkout: skipped, the round's own run: profiles/round6_trained_parity.json).

Trained GDN beta / gamma, PReLU slopes, quantiser levels and entropy weights put activations, the quantiser's tie
rate, the Winograd error and the CDF shapes somewhere else than the seeded random draw of the other tests:
  * 512x1024 and the metric size 2048x4096: symbols vs the oracle's (ties counted), the engine codes the ORACLE's
    symbols into the ORACLE's bytes and decodes them back, HIP synthesis <= 1e-4 from the oracle's image;
  * the reconstruction is an image: viewport PSNR far above the ~6 dB of random weights."""
import json
import os
import sys

import pytest
import torch

from oracle import pconv_cpu as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PACK = os.path.join(ROOT, "trained", "r6", "codec_3_56.pack.pt")


@pytest.fixture(scope="module")
def weights(tmp_path_factory):
    if not os.path.exists(PACK):
        pytest.skip("no trained weights (trained/r6/codec_3_56.pack.pt: run tools/train_round6.py on the GPU box)")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import weights_pack
    d = str(tmp_path_factory.mktemp("trained"))
    weights_pack.unpack(PACK, d)
    return d


def codec(weights_dir):
    from pseudocylindrical_convolution_amd import pseudo_codec as PC
    from pseudocylindrical_convolution_amd.PCONV_operator import backend
    dev = backend.device_of(0)
    enc, dec = PC.PseudoEncoder(56, 0).to(dev).eval(), PC.PseudoDecoder(56, 0).to(dev).eval()
    PC.load_models(enc, weights_dir + "/3_56_encoder.pt", weights_dir + "/3_56_ent.pt", dev)   # pseudo_codec.py:223-227
    PC.load_models(dec, weights_dir + "/3_56_decoder.pt", weights_dir + "/3_56_ent.pt", dev)
    return enc.eval(), dec.eval()


def frame(h, w, seed):
    from pseudocylindrical_convolution_amd.SphereDataset import procedural_erp
    x = procedural_erp(h, w, 424243 + seed, 1.5)
    return ((x * 255.0 + 0.5).clamp_(0, 255).to(torch.uint8).float() / 255.0).unsqueeze(0).contiguous()   # an 8-bit image


def oracle_run(weights_dir, x, H, W, path):
    from pseudocylindrical_convolution_amd.PCONV_operator import backend
    from oracle import coder_cpu
    backend.use(O, coder_cpu)
    O.set_detmath(True)
    threads = torch.get_num_threads()
    torch.set_num_threads(O.set_num_threads())
    try:
        enc, dec = codec(weights_dir)
        with torch.no_grad():
            sym = enc.ent.fill(enc.symbols(x)).clone()
        enc.ent.start(path)
        enc.ent(sym)
        with open(path, "rb") as f:
            data = f.read()
        dec.ent.start(path)
        back = dec.ent(2 * (H // 256), 2 * (W // 16)).clone()
        assert torch.equal(back, sym)
        rec = dec.reconstruct(back).clone()
    finally:
        backend.reset()
        torch.set_num_threads(threads)
    return sym, data, rec


def record(key, value):
    path = os.path.join(ROOT, "gpurun_out", "trained_parity.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[key] = value
        with open(path, "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)
    except (OSError, ValueError):
        pass


@pytest.mark.timeout(1700)
@pytest.mark.parametrize("H,W,max_ties", [(512, 1024, 8), (2048, 4096, 64)])
def test_trained_codec_equals_the_oracle(hip_backend, weights, tmp_path, H, W, max_ties):
    from pseudocylindrical_convolution_amd.engine import CodecEngine
    from pseudocylindrical_convolution_amd.pseudo_codec import ViewportMetrics
    O.set_detmath(True)
    x = frame(H, W, H)
    csym, cbytes, crec = oracle_run(weights, x, H, W, str(tmp_path / "cpu.bin"))
    assert len(csym.unique()) >= 4
    enc, dec = codec(weights)
    eng = CodecEngine(56, 0, enc, dec)
    h2, w2 = 2 * (H // 256), 2 * (W // 16)
    gsym = eng.symbols(x.cuda()).cpu()
    ties = int((gsym != csym).sum())
    assert ties <= max_ties, "%d of %d symbols differ from the oracle's" % (ties, csym.numel())
    streams = eng._engine("enc", h2, w2, 1).encode(csym.cuda().contiguous())
    assert streams[0] == cbytes, "engine stream %d bytes, oracle %d" % (len(streams[0]), len(cbytes))
    if ties == 0:
        assert eng.encode(x.cuda())[0] == cbytes
    back = eng._engine("dec", h2, w2, 1).decode([cbytes]).cpu()
    assert torch.equal(back, csym)
    rec = eng.decode([cbytes], H, W)
    err = (rec.cpu() - crec).abs().max().item()
    psnr, ssim = ViewportMetrics(0)(x.cuda(), rec)
    record("%dx%d" % (H, W), {"quantiser_ties": ties, "symbols": csym.numel(), "bytes": len(cbytes),
                               "bpp": round(len(cbytes) * 8.0 / H / W, 4), "reconstruction_max_abs_err_vs_oracle": err,
                               "viewport_psnr_db": round(float(psnr), 2), "viewport_ssim": round(float(ssim), 4),
                               "symbol_histogram": torch.bincount(csym.flatten().long(), minlength=8).tolist()})
    assert err < 1e-4, "reconstruction differs from the oracle by %g" % err
    assert psnr > 20.0, "a trained model reconstructs an image (viewport PSNR %.1f dB)" % psnr


def test_trained_model_per_op_loops_write_the_engines_file(hip_backend, weights, tmp_path):
    """the reference's op-by-op loops (pseudo_codec.py:97-114, 145-160) on the HIP per-op kernels and the native engine
    write and read the same file with the TRAINED entropy model too (sharp, position-dependent CDFs instead of the
    near-uniform ones of randn * 0.05 weights)"""
    H, W = 512, 1024
    x = frame(H, W, 7).cuda()
    enc, dec = codec(weights)
    a, b = str(tmp_path / "engine.bin"), str(tmp_path / "per_op.bin")
    enc(x, a)
    enc.forward_per_op(x, b)
    with open(a, "rb") as f, open(b, "rb") as g:
        fa, fb = f.read(), g.read()
    assert fa == fb and len(fa) > 1000
    ra = dec(a, H, W).clone()
    rb = dec.forward_per_op(b, H, W)
    assert torch.equal(ra, rb)
