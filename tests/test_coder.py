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


@pytest.mark.skipif(coder_cpu.ref_lib() is None, reason="oracle/_ref not built (reference tree absent)")
def test_product_matches_reference_library_live(tmp_path):
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
