"""GPU: the native entropy engine against the per-op path and the CPU oracle.

The engine drives the same step kernels as the PCONV op classes, so its streams
must be byte-identical to PseudoEncoder's files; frames coded in lock-step must
equal frames coded one by one; decode must return the exact symbols."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _codec():
    from pseudocylindrical_convolution_amd import pseudo_codec as PC
    torch.manual_seed(1234)
    enc, dec = PC.PseudoEncoder(56, 0), PC.PseudoDecoder(56, 0)
    g = torch.Generator().manual_seed(7)
    sd = {k: torch.randn(v.shape, generator=g) * 0.05 for k, v in enc.ent.state_dict().items()}
    enc.ent.load_state_dict(sd)
    dec.ent.load_state_dict(sd)
    dec.quant.weight.data.copy_(enc.quant.weight.data)
    return enc, dec


def _frames(n, h, w, seed=1):
    return torch.rand(n, 3, h, w, generator=torch.Generator().manual_seed(seed)).cuda()


def test_engine_stream_equals_per_op_path(hip_backend, tmp_path):
    from pseudocylindrical_convolution_amd.engine import CodecEngine
    enc, dec = _codec()
    eng = CodecEngine(56, 0, enc, dec)
    x = _frames(1, 256, 512)
    path = str(tmp_path / "ref.bin")
    enc.forward_per_op(x, path)  # the reference's op-by-op loop
    streams = eng.encode(x)
    with open(path, "rb") as f:
        assert streams[0] == f.read()
    # the module's forward (native engine inside) writes the same file
    fast = str(tmp_path / "fast.bin")
    enc(x, fast)
    with open(fast, "rb") as f:
        assert streams[0] == f.read()
    sym = eng.symbols(x)
    out = eng._engine("dec", sym.shape[2], sym.shape[3], 1).decode(streams)
    assert torch.equal(out, sym)
    rec_engine = eng.decode(streams, 256, 512)
    rec_per_op = dec.forward_per_op(path, 256, 512)
    assert torch.equal(rec_engine, rec_per_op)
    assert torch.equal(dec(fast, 256, 512), rec_per_op)


def test_engine_lockstep_batch_equals_single_frames(hip_backend):
    from pseudocylindrical_convolution_amd.engine import CodecEngine
    enc, dec = _codec()
    eng = CodecEngine(56, 0, enc, dec)
    x = _frames(3, 256, 512, seed=5)
    batch = eng.encode(x)
    single = [eng.encode(x[i:i + 1])[0] for i in range(3)]
    assert batch == single
    assert len(set(batch)) == 3
    rec = eng.decode(batch, 256, 512)
    for i in range(3):
        assert torch.equal(rec[i:i + 1], eng.decode([single[i]], 256, 512))


def test_engine_roundtrip_reference_size(hip_backend):
    from pseudocylindrical_convolution_amd.engine import CodecEngine
    enc, dec = _codec()
    eng = CodecEngine(56, 0, enc, dec)
    x = _frames(1, 512, 1024, seed=3)
    streams = eng.encode(x)
    sym = eng.symbols(x)
    out = eng._engine("dec", sym.shape[2], sym.shape[3], 1).decode(streams)
    assert torch.equal(out, sym)
    assert eng._engine("enc", sym.shape[2], sym.shape[3], 1).symbols_per_image == 14 * 4 * 16 * 836 // 8


def test_engine_detects_corrupt_stream(hip_backend):
    from pseudocylindrical_convolution_amd.engine import CodecEngine
    from pseudocylindrical_convolution_amd._native import PconvError
    enc, dec = _codec()
    eng = CodecEngine(56, 0, enc, dec)
    x = _frames(1, 256, 512, seed=9)
    s = bytearray(eng.encode(x)[0])
    sym = eng.symbols(x)
    e = eng._engine("dec", sym.shape[2], sym.shape[3], 1)
    for i in range(8, len(s), 7):
        s[i] ^= 0x5a
    try:
        out = e.decode([bytes(s)])
    except PconvError:
        return  # decoder assertion fired
    assert not torch.equal(out, sym)


def test_engine_full_size_properties(hip_backend):
    """BASELINE size (4096x2048, 1.5 M symbols per frame): size-independent properties.
    decode(encode(x)) returns exactly the coded symbols; two frames coded together
    (two ping-pong groups) give the streams of the frames coded alone; a stream
    decodes the same alone and beside another one; the reconstruction is finite,
    in range and as close to the input as the reconstruction of the exact symbols."""
    from pseudocylindrical_convolution_amd.engine import CodecEngine
    enc, dec = _codec()
    eng = CodecEngine(56, 0, enc, dec)
    H, W = 2048, 4096
    x = _frames(2, H, W, seed=11)
    sym = eng.symbols(x)
    assert tuple(sym.shape) == (32, 14, 16, 512)
    assert eng._engine("enc", 16, 512, 1).symbols_per_image == 14 * 16 * 16 * (836 * 512 // 1024)
    both = eng.encode(x)
    alone = [eng.encode(x[i:i + 1])[0] for i in range(2)]
    assert both == alone
    out2 = eng._engine("dec", 16, 512, 2).decode(both)
    assert torch.equal(out2, sym)
    out1 = eng._engine("dec", 16, 512, 1).decode([alone[1]])
    assert torch.equal(out1, sym[16:])
    # rate sanity: 8-level symbols cost at most 3 bits, and the model must not expand
    for s in both:
        assert 0 < len(s) * 8 < 3.2 * 14 * 16 * 16 * 418
    rec = eng.decode(both, H, W)
    assert tuple(rec.shape) == (2, 3, H, W) and torch.isfinite(rec).all()
    assert torch.equal(rec, torch.cat([dec.reconstruct(sym[:16]), dec.reconstruct(sym[16:])], 0))


def test_queued_chain_equals_host_driven_chain(hip_backend, monkeypatch):
    """the decoder's queued-ahead chain (scatter kernels wait in pinned memory for the host's
    symbols, table kernels announce their rows there) returns exactly what the host-driven
    loop returns, for one frame and for a lock-step batch in two groups"""
    from pseudocylindrical_convolution_amd.engine import CodecEngine
    enc, dec = _codec()
    eng = CodecEngine(56, 0, enc, dec)
    x = _frames(3, 256, 512, seed=13)
    streams = eng.encode(x)
    sym = eng.symbols(x)
    for n in (1, 3):
        e = eng._engine("dec", sym.shape[2], sym.shape[3], n)
        monkeypatch.setenv("PCONV_ENGINE_CHAIN", "queued")   # (the default picks by frames per group)
        queued = e.decode(streams[:n])
        monkeypatch.setenv("PCONV_ENGINE_CHAIN", "host")
        host = e.decode(streams[:n])
        monkeypatch.setenv("PCONV_ENGINE_CHAIN", "queued")
        assert torch.equal(queued, host)
        assert torch.equal(queued, sym[:16 * n])
        again = e.decode(streams[:n])   # flags and counters are back in their initial state
        assert torch.equal(again, queued)
        monkeypatch.delenv("PCONV_ENGINE_CHAIN")
        assert torch.equal(e.decode(streams[:n]), queued)


def test_eight_frames_take_the_host_driven_chain_by_default(hip_backend):
    """groups of four or more frames decode through the host-driven chain (faster there); the
    choice is by frames per group, the result is the same"""
    from pseudocylindrical_convolution_amd.engine import CodecEngine
    enc, dec = _codec()
    eng = CodecEngine(56, 0, enc, dec)
    x = _frames(8, 256, 512, seed=19)
    streams = eng.encode(x)
    sym = eng.symbols(x)
    e = eng._engine("dec", sym.shape[2], sym.shape[3], 8)
    assert torch.equal(e.decode(streams), sym)


def test_pipelined_decode_equals_plain_decode(hip_backend, monkeypatch):
    """CodecEngine.decode with DECODE_CHUNK: the entropy decoder of chunk k+1 runs beside the
    synthesis of chunk k (second stream, host thread, two engines) -- same images"""
    from pseudocylindrical_convolution_amd.engine import CodecEngine
    enc, dec = _codec()
    eng = CodecEngine(56, 0, enc, dec)
    x = _frames(5, 256, 512, seed=17)
    streams = eng.encode(x)
    plain = eng.decode(streams, 256, 512)
    monkeypatch.setattr(CodecEngine, "DECODE_CHUNK", 2)
    piped = eng.decode(streams, 256, 512)
    assert torch.equal(plain, piped)
    # chunked encode == one lock-step encode of all frames == frames one by one
    monkeypatch.setattr(CodecEngine, "ENCODE_CHUNK", 8)
    assert eng.encode(x) == streams
    monkeypatch.setattr(CodecEngine, "ENCODE_CHUNK", 1)
    assert eng.encode(x) == streams


@pytest.mark.parametrize("vd", [112, 192])
def test_engine_wide_models_equal_per_op_path(vd, hip_backend, tmp_path):
    """The other model widths of the reference's tables (valid_dim 112 / 192: 28 / 48 channel groups,
    pseudo_codec.py:18-23): the band kernels' instantiations for 28 / 84 and 48 / 144 input channels (weights
    in several register chunks, group slabs fetched from global memory in the bulk pass) write the stream the
    op-by-op loop writes, and decode it back to the symbols -- 1280 x 512 frames: 10 symbol rows per tile = two
    row chunks of the step kernel (8 + 2), and a two-frame lock-step group."""
    from pseudocylindrical_convolution_amd import pseudo_codec as PC
    from pseudocylindrical_convolution_amd.engine import CodecEngine
    torch.manual_seed(4321)
    enc, dec = PC.PseudoEncoder(vd, 0).eval(), PC.PseudoDecoder(vd, 0).eval()
    g = torch.Generator().manual_seed(17)
    sd = {k: torch.randn(v.shape, generator=g) * 0.03 for k, v in enc.ent.state_dict().items()}
    enc.ent.load_state_dict(sd)
    dec.ent.load_state_dict(sd)
    dec.quant.weight.data.copy_(enc.quant.weight.data)
    eng = CodecEngine(vd, 0, enc, dec)
    x = _frames(2, 1280, 512, seed=13)
    streams = eng.encode(x)
    path = str(tmp_path / "perop.bin")
    enc.forward_per_op(x[:1], path)
    with open(path, "rb") as f:
        assert streams[0] == f.read()
    sym = eng.symbols(x)
    assert tuple(sym.shape) == (32, vd // 4, 10, 64)
    assert torch.equal(eng._engine("dec", 10, 64, 2).decode(streams), sym)
    assert torch.equal(eng._engine("dec", 10, 64, 1).decode(streams[1:]), sym[16:])


def test_engine_buffers_are_zeroed_once_not_per_call(hip_backend):
    """the engine clears its context / activation buffers once per group (csrc/engine.cpp clear()): a second call
    on the same engine runs over the first call's values -- everything it reads through a non-zero weight has been
    rewritten by then.  Frames B after frames A on one engine == frames B on a fresh engine, streams and symbols."""
    from pseudocylindrical_convolution_amd.engine import CodecEngine
    enc, dec = _codec()
    a, b = _frames(2, 512, 1024, seed=31), _frames(2, 512, 1024, seed=32)
    used = CodecEngine(56, 0, enc, dec)
    sa = used.encode(a)
    used.decode(sa, 512, 1024)
    sb = used.encode(b)
    fresh = CodecEngine(56, 0, enc, dec)
    assert sb == fresh.encode(b) and sb != sa
    sym = fresh.symbols(b)
    assert torch.equal(used._engine("dec", sym.shape[2], sym.shape[3], 2).decode(sb), sym)
    assert torch.equal(used.decode(sb, 512, 1024), fresh.decode(sb, 512, 1024))


def test_host_constrained_plan_decodes_the_same_symbols(hip_backend, monkeypatch, tmp_path):
    """A rank with a small share of the host (cgroup quota / LOCAL_WORLD_SIZE below frames + 1: csrc/engine.cpp
    host_plan) decodes with at most one group per CPU, the drivers decoding their groups' frames alone, on the
    host-driven chain: same symbols as the plan of a rank that owns the host, same streams from the encoder."""
    from pseudocylindrical_convolution_amd.engine import EntropyEngine
    enc, _ = _codec()
    ent = enc.ent
    h, w, n = 4, 128, 5
    sym = torch.randint(0, 8, (16 * n, 14, h, w), generator=torch.Generator().manual_seed(23)).float().cuda()
    sym = ent.fill(sym).contiguous()
    for name in ("PCONV_ENGINE_GROUPS", "PCONV_ENGINE_WORKERS", "PCONV_ENGINE_CHAIN", "PCONV_ENGINE_BLOCKING_SYNC",
                 "PCONV_CGROUP_CPU_MAX", "LOCAL_WORLD_SIZE"):
        monkeypatch.delenv(name, raising=False)
    roomy = EntropyEngine(ent, h, w, n, "cuda:0")
    streams = roomy.encode(sym)
    assert torch.equal(roomy.decode(streams), sym)
    quota = tmp_path / "cpu.max"
    quota.write_text("1600000 100000")                       # 16 CPUs ...
    monkeypatch.setenv("PCONV_CGROUP_CPU_MAX", str(quota))
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")              # ... shared by 8 ranks: 2 per rank, 5 frames
    lib = roomy.lib
    assert lib.pconv_ee_wait_mode(roomy.handle) == 0         # a roomy rank keeps the runtime's spinning waits
    import ctypes
    plan = [ctypes.c_int(-1) for _ in range(4)]
    assert lib.pconv_ee_host_plan(n, *[ctypes.addressof(v) for v in plan]) == 0
    assert [v.value for v in plan[:3]] == [min(2, lib.pconv_ee_host_cpus()), 1, 0]
    tight = EntropyEngine(ent, h, w, n, "cuda:0")
    # sleeping waits = blocking EVENTS of this engine (hipEventBlockingSync), no device-wide schedule flag: the plan
    # is fixed at creation and the engine reports it
    assert plan[3].value == 1 and lib.pconv_ee_wait_mode(tight.handle) == 1
    assert tight.encode(sym) == streams
    assert torch.equal(tight.decode(streams), sym)
    # the plan is the engine's, not the environment's: a roomier environment at decode time changes nothing
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "1")
    assert lib.pconv_ee_wait_mode(tight.handle) == 1
    assert torch.equal(tight.decode(streams), sym)
    assert lib.pconv_ee_wait_mode(roomy.handle) == 0
    assert lib.pconv_device_blocking_sync(0) in (0, 1)       # (a query: the explicit opt-in itself is the application's call)
    # one CPU: a single group, one thread
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "16")
    one = EntropyEngine(ent, h, w, n, "cuda:0")
    assert torch.equal(one.decode(streams), sym)


def test_engine_knobs_that_must_not_change_a_stream(hip_backend, monkeypatch):
    """PCONV_ENGINE_ROWS (16-byte packed CDF rows or int32[9] rows + labels across PCIe), the order of the tail
    encode's step ranges (interleaved over the groups or group by group) and their number, the step-by-step
    debugging encoder, the engine's streams on a partition of the compute units, the other matrix-core form of the
    encoder's hidden layers: the same streams, and every decoder configuration returns the coded symbols."""
    from pseudocylindrical_convolution_amd.engine import EntropyEngine
    enc, _ = _codec()
    ent = enc.ent
    h, w, n = 4, 128, 3
    sym = torch.randint(0, 8, (16 * n, 14, h, w), generator=torch.Generator().manual_seed(29)).float().cuda()
    sym = ent.fill(sym).contiguous()
    for name in ("PCONV_ENGINE_ROWS", "PCONV_ENGINE_ENCODE_INTERLEAVE", "PCONV_ENGINE_ENCODE_RANGES",
                 "PCONV_ENGINE_STEPWISE_ENCODER", "PCONV_ENGINE_CHAIN", "PCONV_ENGINE_CU_MASK", "PCONV_EE_MFMA_FORM",
                 "PCONV_EE_FUSE_TABLES", "PCONV_EE_FUSE_PPW"):
        monkeypatch.delenv(name, raising=False)
    ref_engine = EntropyEngine(ent, h, w, n, "cuda:0")
    ref = ref_engine.encode(sym)
    for env in ({"PCONV_ENGINE_ROWS": "int32"}, {"PCONV_ENGINE_ENCODE_INTERLEAVE": "0"},
                {"PCONV_ENGINE_ENCODE_RANGES": "1"}, {"PCONV_ENGINE_ENCODE_RANGES": "7"},
                {"PCONV_ENGINE_ROWS": "int32", "PCONV_ENGINE_CHAIN": "host"}, {"PCONV_ENGINE_CHAIN": "queued"},
                {"PCONV_ENGINE_STEPWISE_ENCODER": "1"}, {"PCONV_ENGINE_CU_MASK": "0:64"},
                {"PCONV_EE_MFMA_FORM": "16x4"},
                # (r6) the decoder's last layer + table kernel as one launch, on both chains
                {"PCONV_EE_FUSE_TABLES": "1", "PCONV_ENGINE_CHAIN": "host"},
                {"PCONV_EE_FUSE_TABLES": "1", "PCONV_ENGINE_CHAIN": "queued"},
                {"PCONV_EE_FUSE_TABLES": "1", "PCONV_EE_FUSE_PPW": "2"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        e = EntropyEngine(ent, h, w, n, "cuda:0")
        assert e.encode(sym) == ref, env
        assert torch.equal(e.decode(ref), sym), env
        for k in env:
            monkeypatch.delenv(k)
    # the per-engine setter wins over the environment and is reset by a plain encode()
    ref_engine.encode_begin(sym, ranges=5)
    assert ref_engine.encode_end() == ref
