#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the parts of the REFERENCE that can run in the
authoring container:

  coder_*.npz     bitstreams written by the reference arithmetic coder, compiled
                  from its own sources into oracle/_ref/libcoder_ref.so
                  (oracle/Makefile), for seeded (tables, symbols)
  set_weight.npz  PCONV_operator/base.py:set_weight outputs (imported by path)
  torch_helpers.npz  GDN.LowerBound, pytorch_ssim.SSIM(11,3), StubMask.Extract
                  outputs on seeded inputs (pure-torch files imported by path)

The fixtures are data (inputs + expected outputs); no reference source is copied.
Run from the repo root:  python tests/golden/gen_golden.py
"""
import importlib.util
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def random_tables(rng, n, skew):
    w = (rng.gamma(0.3, 1.0, size=(n, 8)) + 1e-6) if skew else (rng.random((n, 8)) + 0.01)
    w = w / w.sum(1, keepdims=True)
    c = np.floor(np.cumsum(w, 1) * (65536 - 8)).astype(np.int64) + np.arange(1, 9)
    t = np.concatenate([np.zeros((n, 1), np.int64), c], 1)
    t[:, 8] = 65536
    assert (np.diff(t, axis=1) > 0).all()
    return t.astype(np.int32)


def coder_fixtures():
    from oracle import coder_cpu
    assert coder_cpu.ref_lib() is not None, "build oracle/_ref first: make -C oracle ref"
    rng = np.random.default_rng(0)
    tmp = tempfile.mkdtemp()
    cases = {
        # BASELINE config #1: 32x32 symbols, uniform 9-entry CDF
        "uniform_32x32": (np.tile(np.arange(9, dtype=np.int32) * 8192, (1024, 1)),
                          torch.randint(0, 8, (1024,), generator=torch.Generator().manual_seed(0)).numpy().astype(np.int32)),
        "random_32x32": None, "skewed_4096": None, "single": None, "empty": None,
    }
    t = random_tables(rng, 1024, False)
    cases["random_32x32"] = (t, (rng.integers(0, 65536, 1024)[:, None] >= t[:, 1:]).sum(1).astype(np.int32))
    t = random_tables(rng, 4096, True)
    cases["skewed_4096"] = (t, (rng.integers(0, 65536, 4096)[:, None] >= t[:, 1:]).sum(1).astype(np.int32))
    t = random_tables(rng, 1, False)
    cases["single"] = (t, np.array([5], np.int32))
    cases["empty"] = (np.zeros((0, 9), np.int32), np.zeros((0,), np.int32))
    for name, (tab, sym) in cases.items():
        path = os.path.join(tmp, name)
        c = coder_cpu.RefCoder(path)
        c.start_encoder()
        if len(sym):
            c.encodes(torch.from_numpy(tab), 8, torch.from_numpy(sym), len(sym))
        c.end_encoder()
        with open(path, "rb") as f:
            stream = np.frombuffer(f.read(), dtype=np.uint8)
        np.savez_compressed(os.path.join(OUT, "coder_%s.npz" % name), tables=tab, symbols=sym, stream=stream)
        print("coder", name, len(sym), "symbols ->", len(stream), "bytes")


def weight_fixtures():
    base = load(os.path.join(REF, "PCONV_operator", "base.py"), "ref_base")
    cases = [(16, False, False), (16, True, False), (8, True, True), (32, True, False), (4, False, False), (32, False, True)]
    np.savez(os.path.join(OUT, "set_weight.npz"),
             args=np.array(cases, dtype=np.int32),
             **{"out_%d" % i: np.array(base.set_weight(n, bool(o), bool(m)), dtype=np.float64) for i, (n, o, m) in enumerate(cases)})
    print("set_weight", len(cases), "cases")


def torch_fixtures():
    gdn = load(os.path.join(REF, "PCONV_operator", "GDN.py"), "ref_gdn")
    ssim = load(os.path.join(REF, "PCONV_operator", "pytorch_ssim.py"), "ref_ssim")
    stub = load(os.path.join(REF, "PCONV_operator", "StubMask.py"), "ref_stub")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 7, generator=g)
    bound = torch.FloatTensor([0.25])
    lb = gdn.LowerBound.apply(x, bound)
    a = torch.rand(2, 3, 40, 48, generator=g)
    b = (a + 0.1 * torch.randn(2, 3, 40, 48, generator=g)).clamp(0, 1)
    s = ssim.SSIM(11, 3)(a, b)
    e = torch.randn(2, 9, 3, 4, generator=g)
    ex = stub.Extract(5)(e)
    np.savez_compressed(os.path.join(OUT, "torch_helpers.npz"), lb_x=x.numpy(), lb_bound=bound.numpy(), lb_out=lb.numpy(),
                        ssim_a=a.numpy(), ssim_b=b.numpy(), ssim_out=np.array(s.item()), ex_in=e.numpy(), ex_out=ex.numpy())
    print("torch helpers: LowerBound, SSIM=%.6f, Extract" % s.item())


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    coder_fixtures()
    weight_fixtures()
    torch_fixtures()
