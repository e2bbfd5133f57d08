"""Arithmetic coder: product (libpconv_coder.so) vs the golden streams written by
the REFERENCE coder (tests/golden/coder_*.npz, tests/golden/gen_golden.py), vs the
bit-at-a-time oracle restatement, and -- when oracle/_ref is built -- vs the
reference library live.  BASELINE config #1 is `uniform_32x32`."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import coder_cpu
from pseudocylindrical_convolution_amd import coder as product

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = sorted(glob.glob(os.path.join(GOLD, "coder_*.npz")))


def _encode(cls, path, tab, sym):
    c = cls(path)
    c.start_encoder()
    if len(sym):
        c.encodes(torch.from_numpy(tab), 8, torch.from_numpy(sym), len(sym))
    c.end_encoder()
    with open(path, "rb") as f:
        return f.read()


def _decode(cls, path, tab, n):
    c = cls(path)
    c.start_decoder()
    out = c.decodes(torch.from_numpy(tab), 8, n)
    return out.numpy()[:n].astype(np.int32)


@pytest.mark.parametrize("case", CASES, ids=[os.path.basename(c)[6:-4] for c in CASES])
def test_product_matches_reference_golden_stream(case, tmp_path):
    d = np.load(case)
    tab, sym, gold = d["tables"], d["symbols"], d["stream"].tobytes()
    got = _encode(product.coder, str(tmp_path / "p.bin"), tab, sym)
    assert got == gold
    with open(str(tmp_path / "g.bin"), "wb") as f:
        f.write(gold)
    assert (_decode(product.coder, str(tmp_path / "g.bin"), tab, len(sym)) == sym).all()


@pytest.mark.parametrize("case", [c for c in CASES if "4096" not in c], ids=lambda c: os.path.basename(c)[6:-4])
def test_oracle_restatement_matches_reference_golden_stream(case, tmp_path):
    d = np.load(case)
    tab, sym, gold = d["tables"], d["symbols"], d["stream"].tobytes()
    assert _encode(coder_cpu.PyCoder, str(tmp_path / "o.bin"), tab, sym) == gold
    with open(str(tmp_path / "g.bin"), "wb") as f:
        f.write(gold)
    assert (_decode(coder_cpu.PyCoder, str(tmp_path / "g.bin"), tab, len(sym)) == sym).all()


def test_product_matches_reference_library_live(tmp_path):
    # decided at RUN time (ref_lib builds oracle/_ref when the reference tree is there): a collection-time
    # skipif ran before anything had built it and skipped silently on a fresh tree
    if coder_cpu.ref_lib() is None:
        pytest.skip("oracle/_ref not built and the reference tree is absent (GPU box)")
    rng = np.random.default_rng(42)
    for n, skew in [(1, False), (7, True), (5000, True), (5000, False)]:
        w = (rng.gamma(0.2, 1.0, size=(n, 8)) + 1e-7) if skew else (rng.random((n, 8)) + 0.01)
        w = w / w.sum(1, keepdims=True)
        c = np.floor(np.cumsum(w, 1) * (65536 - 8)).astype(np.int64) + np.arange(1, 9)
        tab = np.concatenate([np.zeros((n, 1), np.int64), c], 1).astype(np.int32)
        tab[:, 8] = 65536
        sym = (rng.integers(0, 65536, n)[:, None] >= tab[:, 1:]).sum(1).astype(np.int32)
        ref = _encode(coder_cpu.RefCoder, str(tmp_path / "r.bin"), tab, sym)
        got = _encode(product.coder, str(tmp_path / "p.bin"), tab, sym)
        assert got == ref
        assert (_decode(coder_cpu.RefCoder, str(tmp_path / "p.bin"), tab, n) == sym).all()


def test_per_step_calls_concatenate_like_one_call(tmp_path):
    d = np.load(os.path.join(GOLD, "coder_random_32x32.npz"))
    tab, sym = d["tables"], d["symbols"]
    c = product.coder(str(tmp_path / "s.bin"))
    c.start_encoder()
    for lo in range(0, 1024, 100):
        hi = min(lo + 100, 1024)
        c.encodes(torch.from_numpy(tab[lo:hi].copy()), 8, torch.from_numpy(sym[lo:hi].copy()), hi - lo)
    c.end_encoder()
    assert c.bytes() == d["stream"].tobytes()
    c.start_decoder()
    got = []
    for lo in range(0, 1024, 37):
        hi = min(lo + 37, 1024)
        got.append(c.decodes(torch.from_numpy(tab[lo:hi].copy()), 8, hi - lo).numpy()[:hi - lo])
    assert (np.concatenate(got).astype(np.int32) == sym).all()


def test_single_symbol_api(tmp_path):
    tab = torch.tensor([0, 10, 20, 65536], dtype=torch.int32)
    c = product.coder(str(tmp_path / "one.bin"))
    c.start_encoder()
    for s in (2, 0, 1, 2, 2):
        c.encode(tab, 3, 65536, s)
    c.end_encoder()
    c.start_decoder()
    assert [c.decode(tab, 3, 65536) for _ in range(5)] == [2, 0, 1, 2, 2]


def test_errors_where_the_reference_throws(tmp_path):
    c = product.coder(str(tmp_path / "e.bin"))
    c.start_encoder()
    flat = torch.tensor([[0, 5, 5, 65536]], dtype=torch.int32)  # symbol 1 has zero frequency
    with pytest.raises(product.CoderError, match="zero frequency"):
        c.encodes(flat, 3, torch.tensor([1], dtype=torch.int32), 1)
    c2 = product.coder(str(tmp_path / "e2.bin"))
    with pytest.raises(product.CoderError):
        c2.encodes(flat, 3, torch.tensor([0], dtype=torch.int32), 1)  # encoder not started
    c3 = product.coder(str(tmp_path / "e3.bin"))
    c3.start_encoder()
    big = torch.tensor([[0, 1, 2 ** 31 - 1]], dtype=torch.int32)       # total above MAX_TOTAL
    with pytest.raises(product.CoderError, match="total is too large"):
        c3.encodes(big, 2, torch.tensor([0], dtype=torch.int32), 1)
    with pytest.raises(product.CoderError):
        c3.encodes(flat.float(), 3, torch.tensor([0], dtype=torch.int32), 1)  # wrong dtype


def test_missing_file_decodes_as_zero_stream(tmp_path):
    # an unopened ifstream reads EOF -> zero bits (ArithmeticCoder.cpp:121-126)
    c = product.coder(str(tmp_path / "absent.bin"))
    c.start_decoder()
    tab = torch.tensor([[0, 8192, 65536]], dtype=torch.int32)
    assert int(c.decodes(tab, 2, 1)[0]) == 0


def _rows8(rng, n):
    w = rng.gamma(0.3, 1.0, size=(n, 8)) + 1e-6
    w = w / w.sum(1, keepdims=True)
    c = np.floor(np.cumsum(w, 1) * (65536 - 8)).astype(np.int64) + np.arange(1, 9)
    tab = np.concatenate([np.zeros((n, 1), np.int64), c], 1).astype(np.int32)
    tab[:, 8] = 65536
    sym = (rng.integers(0, 65536, n)[:, None] >= tab[:, 1:]).sum(1).astype(np.int32)
    return tab, sym


def test_rows_of_another_total_inside_a_batch(tmp_path):
    """The 8-symbol / total-65536 rows have a loop of their own in the product (coder.cpp: the interval lives in
    registers there); a row of another total ends it and the general path goes on.  Same bytes as the bit-at-a-time
    restatement, and the stream decodes."""
    rng = np.random.default_rng(5)
    tab, sym = _rows8(rng, 600)
    for r in (0, 1, 299, 300, 599):                      # rows whose total is 40000, not 65536
        tab[r] = (tab[r].astype(np.int64) * 40000 // 65536).astype(np.int32)
        tab[r, 1:] = np.maximum(tab[r, 1:], tab[r, :-1] + 1)
        tab[r, 8] = max(int(tab[r, 8]), 40000)
    want = _encode(coder_cpu.PyCoder, str(tmp_path / "o.bin"), tab, sym)
    assert _encode(product.coder, str(tmp_path / "p.bin"), tab, sym) == want
    assert (_decode(product.coder, str(tmp_path / "p.bin"), tab, len(sym)) == sym).all()


def test_error_inside_a_batch_leaves_the_rows_before_it_coded(tmp_path):
    rng = np.random.default_rng(6)
    tab, sym = _rows8(rng, 200)
    bad = tab.copy()
    bad[120, 4] = bad[120, 3]                            # symbol 3 of row 120 has zero frequency
    sym[120] = 3
    c = product.coder(str(tmp_path / "z.bin"))
    c.start_encoder()
    with pytest.raises(product.CoderError, match="zero frequency"):
        c.encodes(torch.from_numpy(bad), 8, torch.from_numpy(sym), 200)
    # the coder's state is that after rows 0..119: go on with good rows and compare with a clean run
    c.encodes(torch.from_numpy(tab[120:].copy()), 8, torch.from_numpy(sym[120:].copy()), 80)
    c.end_encoder()
    assert c.bytes() == _encode(product.coder, str(tmp_path / "g.bin"), tab, sym)
    c2 = product.coder(str(tmp_path / "r.bin"))
    c2.start_encoder()
    sym2 = sym.copy()
    sym2[50] = 8                                         # not a symbol of an 8-symbol row
    with pytest.raises(product.CoderError, match="out of range"):
        c2.encodes(torch.from_numpy(tab), 8, torch.from_numpy(sym2), 200)


def test_decoder_reports_zero_frequency_and_survives_garbage(tmp_path):
    rng = np.random.default_rng(7)
    tab, sym = _rows8(rng, 300)
    noise = rng.integers(0, 256, 4000, dtype=np.uint8).tobytes()
    with open(str(tmp_path / "n.bin"), "wb") as f:
        f.write(noise)
    # any bit string is a valid arithmetic code of SOME symbols: product and restatement must name the same ones
    got = _decode(product.coder, str(tmp_path / "n.bin"), tab, 300)
    assert (got == _decode(coder_cpu.PyCoder, str(tmp_path / "n.bin"), tab, 300)).all()
    assert _encode(product.coder, str(tmp_path / "b.bin"), tab, got)[:8] == noise[:8]


def _pack_rows16(tab, sym=None):
    """int32[n, 9] CDF rows (+ symbols) -> the engine's packed rows uint16[n, 8] (include/pconv_coder.h)"""
    tab = np.asarray(tab, dtype=np.int64)
    rows = np.zeros((len(tab), 8), dtype=np.uint16)
    rows[:, :7] = (tab[:, 1:8] & 0xffff).astype(np.uint16)
    aux = np.zeros(len(tab), dtype=np.uint32)
    for k in range(1, 8):
        aux |= (tab[:, k] == 65536).astype(np.uint32) << (7 + k)
    if sym is not None:
        aux |= np.asarray(sym, dtype=np.uint32) & 0xff
    rows[:, 7] = aux.astype(np.uint16)
    return np.ascontiguousarray(rows)


@pytest.mark.parametrize("case", CASES, ids=[os.path.basename(c)[6:-4] for c in CASES])
def test_packed_rows_write_and_read_the_reference_golden_stream(case):
    """pconv_coder_encodes_rows16 / _decodes_rows16_i32 (16 bytes per symbol across PCIe instead of 40) against the
    streams of the REFERENCE coder: same bytes, same symbols"""
    from pseudocylindrical_convolution_amd import _native
    lib = _native.coder_lib()
    d = np.load(case)
    tab, sym, gold = d["tables"], d["symbols"], d["stream"].tobytes()
    rows = _pack_rows16(tab, sym)
    c = lib.pconv_coder_new(None)
    try:
        assert lib.pconv_coder_start_encoder(c) == 0
        half = len(sym) // 2                         # two calls: the state carries over like the int32 loop's
        assert lib.pconv_coder_encodes_rows16(c, rows.ctypes.data, half) == 0
        assert lib.pconv_coder_encodes_rows16(c, rows[half:].ctypes.data, len(sym) - half) == 0
        assert lib.pconv_coder_end_encoder(c) == 0
        n = ctypes_size()
        p = lib.pconv_coder_bytes(c, n)
        got = bytes(bytearray(p[i] for i in range(n._obj.value))) if n._obj.value else b""
        assert got == gold
        buf = np.frombuffer(gold, dtype=np.uint8).copy()
        assert lib.pconv_coder_start_decoder_mem(c, buf.ctypes.data, len(buf)) == 0
        out = np.zeros(max(len(sym), 1), dtype=np.int32)
        assert lib.pconv_coder_decodes_rows16_i32(c, _pack_rows16(tab).ctypes.data, out.ctypes.data, len(sym)) == 0
        assert (out[:len(sym)] == sym).all()
    finally:
        lib.pconv_coder_free(c)


def ctypes_size():
    import ctypes
    return ctypes.byref(ctypes.c_size_t(0))


def test_packed_rows_keep_entries_of_65536_exact():
    """a row whose top bins are empty (c_k == 65536 for k < 8) is stored with a flag bit, not wrapped to 0: coding
    a live symbol gives the int32 loop's bytes, coding an empty bin the same "zero frequency" error"""
    from pseudocylindrical_convolution_amd import _native
    lib = _native.coder_lib()
    tab = np.array([[0, 100, 30000, 65000, 65536, 65536, 65536, 65536, 65536]] * 64, dtype=np.int32)
    sym = (np.arange(64) % 4).astype(np.int32)
    a, b = lib.pconv_coder_new(None), lib.pconv_coder_new(None)
    try:
        for c in (a, b):
            assert lib.pconv_coder_start_encoder(c) == 0
        assert lib.pconv_coder_encodes(a, tab.ctypes.data, 8, sym.ctypes.data, 64) == 0
        assert lib.pconv_coder_encodes_rows16(b, _pack_rows16(tab, sym).ctypes.data, 64) == 0
        out = []
        for c in (a, b):
            assert lib.pconv_coder_end_encoder(c) == 0
            n = ctypes_size()
            p = lib.pconv_coder_bytes(c, n)
            out.append(bytes(bytearray(p[i] for i in range(n._obj.value))))
        assert out[0] == out[1] and len(out[0]) > 4
        bad = np.array([5], dtype=np.int32)      # an empty bin
        assert lib.pconv_coder_start_encoder(a) == 0 and lib.pconv_coder_start_encoder(b) == 0
        ra = lib.pconv_coder_encodes(a, tab.ctypes.data, 8, bad.ctypes.data, 1)
        rb = lib.pconv_coder_encodes_rows16(b, _pack_rows16(tab[:1], bad).ctypes.data, 1)
        assert ra == rb and ra < 0
    finally:
        lib.pconv_coder_free(a)
        lib.pconv_coder_free(b)
