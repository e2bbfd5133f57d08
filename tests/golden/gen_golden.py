#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the parts of the REFERENCE that can run in the
authoring container:

  coder_*.npz     bitstreams written by the reference arithmetic coder, compiled
                  from its own sources into oracle/_ref/libcoder_ref.so
                  (oracle/Makefile), for seeded (tables, symbols)
  set_weight.npz  PCONV_operator/base.py:set_weight outputs (imported by path)
  torch_helpers.npz  GDN.LowerBound, pytorch_ssim.SSIM(11,3), StubMask.Extract
                  outputs on seeded inputs (pure-torch files imported by path)

  reference_graph.npz  the reference's OWN model_zoo_v2.py (EncoderV2, DecoderV2,
                  CMPNetV2MF; imported by path, unchanged) run on this repo's drop-in
                  modules with the CPU oracle underneath: state_dict key -> shape maps,
                  a checksum per seeded parameter, and the outputs on seeded inputs

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


# ---- the reference graph on the drop-in (model_zoo_v2.py:129-211, 322-334) ------------------
GRAPH_SEED = 1234          # torch.manual_seed before each model is constructed
GRAPH_INPUT_SEED = 5
GRAPH_SIZES = {"small": (256, 512), "ref": (512, 1024)}   # (H, W); "ref" = the reference codec's own frame size
GRAPH_STRIDE = 37          # outputs larger than 64k elements are stored as every 37th element + checksums


def graph_models(zoo, operators, which):
    """the three graphs, each built right after torch.manual_seed(GRAPH_SEED)"""
    torch.manual_seed(GRAPH_SEED)
    if which == "CMPNetV2MF":
        return zoo.CMPNetV2MF(56, 192, 192, 16, 8, True, False, 0)
    ctx = operators.PseudoContextV2(16, True, device=0)
    cls = zoo.EncoderV2 if which == "EncoderV2" else zoo.DecoderV2
    return cls(192, 192, 16, ctx, 0)


def graph_inputs(which, size):
    H, W = GRAPH_SIZES[size]
    g = torch.Generator().manual_seed(GRAPH_INPUT_SEED)
    if which == "EncoderV2":
        return torch.rand(16, 3, H // 16, W, generator=g)
    if which == "DecoderV2":
        return torch.rand(16, 192, H // 256, W // 16, generator=g)
    return torch.rand(1, 3, H, W, generator=g)


def graph_record(t):
    """what the fixture keeps of an output tensor"""
    flat = t.detach().reshape(-1).to(torch.float32)
    d = flat.double()
    keep = flat if flat.numel() <= 65536 else flat[::GRAPH_STRIDE]
    return {"shape": np.array(t.shape, np.int64), "values": keep.numpy().copy(),
            "sum": np.array(d.sum().item()), "abs_sum": np.array(d.abs().sum().item()),
            "sq_sum": np.array((d * d).sum().item())}


GRAPH_CASES = [("EncoderV2", "small"), ("DecoderV2", "small"), ("CMPNetV2MF", "small"),
               ("EncoderV2", "ref"), ("DecoderV2", "ref"), ("CMPNetV2MF", "ref")]


def graph_fixtures():
    import pseudocylindrical_convolution_amd as pkg
    from pseudocylindrical_convolution_amd.PCONV_operator import backend
    from oracle import pconv_cpu, coder_cpu
    backend.use(pconv_cpu, coder_cpu)
    pconv_cpu.set_detmath(True)
    pkg.install_dropin()
    ref_zoo = load(os.path.join(REF, "model_zoo_v2.py"), "ref_model_zoo_v2")   # its imports resolve to the drop-in
    import PCONV_operator as operators
    out = {}
    try:
        for which in ("EncoderV2", "DecoderV2", "CMPNetV2MF"):
            sd = graph_models(ref_zoo, operators, which).state_dict()
            out["%s/keys" % which] = np.array(list(sd))
            out["%s/shapes" % which] = np.array([list(v.shape) + [-1] * (5 - v.dim()) for v in sd.values()], np.int64)
            out["%s/param_sums" % which] = np.array([v.double().sum().item() for v in sd.values()])
        for which, size in GRAPH_CASES:
            net = graph_models(ref_zoo, operators, which).eval()
            with torch.no_grad():
                res = net(graph_inputs(which, size))
            res = res if isinstance(res, tuple) else (res,)
            for i, t in enumerate(res):
                for k, v in graph_record(t).items():
                    out["%s/%s/out%d/%s" % (which, size, i, k)] = v
            print("graph", which, size, [tuple(t.shape) for t in res])
    finally:
        backend.reset()
    np.savez_compressed(os.path.join(OUT, "reference_graph.npz"), **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if "--graph-only" not in sys.argv:
        coder_fixtures()
        weight_fixtures()
        torch_fixtures()
    graph_fixtures()
