"""Host logic of the operator layer, model graph and codec driver, run end to end
on the CPU oracle backend (property (i): encoder -> file -> decoder returns the
exact symbols), plus checkpoint-name compatibility with the reference."""
import os

import numpy as np
import pytest
import torch


def _codec(vd=56):
    from pseudocylindrical_convolution_amd import pseudo_codec as PC
    torch.manual_seed(1234)
    enc, dec = PC.PseudoEncoder(vd, 0), PC.PseudoDecoder(vd, 0)
    g = torch.Generator().manual_seed(7)
    sd = {k: torch.randn(v.shape, generator=g) * 0.05 for k, v in enc.ent.state_dict().items()}
    enc.ent.load_state_dict(sd)
    dec.ent.load_state_dict(sd)
    dec.quant.weight.data.copy_(enc.quant.weight.data)
    return enc, dec


def test_encode_decode_roundtrip_on_the_oracle(oracle_backend, tmp_path):
    enc, dec = _codec()
    x = torch.rand(1, 3, 256, 512, generator=torch.Generator().manual_seed(1))
    path = str(tmp_path / "code.bin")
    sym = enc.symbols(x)
    assert sym.shape == (16, 14, 2, 64)
    enc(x, path)
    assert 1000 < os.path.getsize(path) < 20000
    dec.ent.start(path)
    out = dec.ent(sym.shape[2], sym.shape[3])
    assert torch.equal(out, enc.ent.fill(sym.clone()))          # (i) exact symbol round trip
    rec = dec(path, 256, 512)
    assert rec.shape == (1, 3, 256, 512) and torch.isfinite(rec).all()
    # container file: 16-byte header + the SAME payload, decodes with no size arguments
    from pseudocylindrical_convolution_amd import container
    boxed = str(tmp_path / "code.pcv")
    enc(x, boxed, {"model_idx": 3, "ssim": True})
    head, payload = container.read(boxed)
    with open(path, "rb") as f:
        assert payload == f.read()
    assert head == {"height": 256, "width": 512, "model_idx": 3, "ssim": True, "valid_dim": 56}
    assert os.path.getsize(boxed) == os.path.getsize(path) + container.HEADER_BYTES
    assert torch.equal(dec(boxed, raw=False), rec)
    with pytest.raises(container.ContainerError):
        dec(path, raw=False)                                       # a raw stream has no magic


def test_container_header_fields_and_errors():
    from pseudocylindrical_convolution_amd import container as C
    blob = C.pack(b"\x01\x02\x03", height=2048, width=4096, model_idx=8, ssim=False, valid_dim=192)
    assert len(blob) == 19 and blob[:4] == b"PCVC" and blob[4] == 1
    head, payload = C.unpack(blob)
    assert payload == b"\x01\x02\x03"
    assert head == {"height": 2048, "width": 4096, "model_idx": 8, "ssim": False, "valid_dim": 192}
    assert C.unpack(C.pack(b"", height=256, width=16, model_idx=0, ssim=True, valid_dim=56))[1] == b""
    for bad in (blob[:10], b"XXXX" + blob[4:], blob + b"\x00", blob[:4] + b"\x07" + blob[5:]):
        with pytest.raises(C.ContainerError):
            C.unpack(bad)
    with pytest.raises(C.ContainerError):
        C.pack(b"", height=250, width=512, model_idx=0, ssim=True, valid_dim=56)
    with pytest.raises(C.ContainerError):
        C.pack(b"", height=256, width=512, model_idx=0, ssim=True, valid_dim=58)


def test_state_dict_names_follow_the_reference(oracle_backend):
    enc, dec = _codec()
    ek, dk = set(enc.state_dict()), set(dec.state_dict())
    # parameter names / shapes are part of the checkpoint format (pseudo_codec.py:223-227)
    for k, shape in {
        "encoder.net.0.conv1.weight": (192, 3, 3, 3), "encoder.net.0.relu2.beta": (192,),
        "encoder.net.0.relu2.gamma": (192, 192), "encoder.net.0.short_cut.weight": (192, 3, 1, 1),
        "encoder.net.3.trunk.0.conv2.weight": (96, 96, 3, 3), "encoder.net.3.attention.3.weight": (192, 192, 1, 1),
        "encoder.net.9.weight": (192, 192, 1, 1), "quant.weight": (192, 8), "quant.count": (192, 8),
        "ent.net.0.conv.weight": (3, 42, 14, 5, 5), "ent.net.0.conv.relu": (3, 42),
        "ent.net.1.conv1.conv.weight": (3, 42, 42, 5, 5), "ent.net.6.conv.bias": (3, 42),
    }.items():
        assert k in ek, k
        assert tuple(enc.state_dict()[k].shape) == shape, k
    assert "ent.net.6.conv.relu" not in ek                        # last layer has no PReLU
    for k, shape in {
        "decoder.net.0.conv.weight": (192, 192, 1, 1), "decoder.net.3.conv1.weight": (768, 192, 3, 3),
        "decoder.net.3.relu1.weight": (768,), "decoder.net.3.relu2.gamma": (192, 192),
        "decoder.net.11.weight": (12, 192, 3, 3), "quant.weight": (192, 8),
    }.items():
        assert k in dk, k
        assert tuple(dec.state_dict()[k].shape) == shape, k
    # the complete key -> shape maps are pinned to the reference's own graph in
    # tests/test_reference_graph.py; the codec modules carry exactly those transforms
    fix = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_graph.npz"))
    assert sorted(k[len("encoder."):] for k in ek if k.startswith("encoder.")) == sorted(str(k) for k in fix["EncoderV2/keys"])
    assert sorted(k[len("decoder."):] for k in dk if k.startswith("decoder.")) == sorted(str(k) for k in fix["DecoderV2/keys"])
    # strict round trip through a file, as load_models does
    enc.load_state_dict(dict(enc.state_dict()))


def test_latent_shape_and_size_checks():
    from pseudocylindrical_convolution_amd import pseudo_codec as PC
    assert PC.latent_shape(512, 1024) == (2, 64)
    assert PC.latent_shape(2048, 4096) == (8, 256)
    with pytest.raises(ValueError):
        PC.latent_shape(500, 1024)
    with pytest.raises(ValueError):
        PC.latent_shape(512, 1000)


def test_quantiser_level_update_matches_oracle_semantics(oracle_backend):
    """the reference's driver never calls .eval(): every 100th training-mode call
    merges unused levels (pseudo_quant_cuda.cu:97-143)"""
    from oracle import pconv_cpu as O
    ctx = O.PseudoContextOp(16, 20, [15., 31., 54., 63., 63., 64., 64., 64., 64., 64., 64., 63., 63., 54., 31., 15.])
    op = O.PseudoQuantOp(4, 8, 16, 0.9, 2, 2, 0.1, ctx.addr())
    w = torch.zeros(4, 8)
    w[:, 0] = 0.1
    w[:, 1:] = -2.0
    cnt = torch.zeros(4, 8)
    cnt[:, :5] = 1.0                                               # levels 5..7 unused
    x = torch.rand(16, 4, 2, 64)
    for _ in range(3):
        op.forward(x, w, cnt, True)
    assert torch.allclose(w[:, 4:], torch.full((4, 4), -2.0 - float(torch.log(torch.tensor(4.0)))))
    assert torch.allclose(cnt[:, :5], torch.full((4, 5), 0.9))


def test_split_transforms_equal_frame_by_frame(oracle_backend):
    """CodecEngine runs the small-scale tail of the analysis transform (and the head of the synthesis
    transform) on all frames of a call at once: same bits as the whole transform frame by frame"""
    from pseudocylindrical_convolution_amd.engine import CodecEngine
    enc, dec = _codec()
    eng = CodecEngine(56, 0, enc, dec)
    assert 0 < eng.ANALYSIS_SPLIT < len(enc.encoder.net) and 0 < eng.SYNTHESIS_SPLIT < len(dec.decoder.net) - 2
    x = torch.rand(3, 3, 256, 512, generator=torch.Generator().manual_seed(9))
    one_by_one = torch.cat([enc.ent.fill(enc.symbols(x[i:i + 1])).clone() for i in range(3)], 0)
    sym = eng.symbols(x)
    assert torch.equal(sym, one_by_one)
    rec = eng.reconstruct(sym, 3)
    for i in range(3):
        assert torch.equal(rec[i:i + 1], dec.reconstruct(sym[i * 16:(i + 1) * 16]))
    # every split point gives the same transform
    t = enc.slice(x[:1])
    whole = enc.encoder(t).clone()
    for k in (1, 4, 9):
        assert torch.equal(enc.encoder.forward_range(enc.encoder.forward_range(t, 0, k), k, 10), whole)
    # a split point at or behind the fused tail is an empty second range: the final conv must not run twice
    assert torch.equal(enc.encoder.forward_range(enc.encoder.forward_range(t, 0, 10), 10, 10), whole)
    assert torch.equal(enc.encoder.forward_range(enc.encoder.forward_range(t, 0, 12), 12, 10), whole)
    # ... and CodecEngine reads out-of-range knobs (PCONV_ANALYSIS_SPLIT / PCONV_SYNTHESIS_SPLIT) as 'no split'
    eng.ANALYSIS_SPLIT, eng.SYNTHESIS_SPLIT = 10, 12
    try:
        assert torch.equal(eng.symbols(x), one_by_one)
        rec2 = eng.reconstruct(sym, 3)
        assert torch.equal(rec2, rec)
        eng.ANALYSIS_SPLIT, eng.SYNTHESIS_SPLIT = 9, 11
        assert torch.equal(eng.symbols(x), one_by_one) and torch.equal(eng.reconstruct(sym, 3), rec)
    finally:
        del eng.ANALYSIS_SPLIT, eng.SYNTHESIS_SPLIT
