"""The reference's OWN operator package (`/root/reference/PCONV_operator`, imported unchanged)
on top of this repo's native modules: every wrapper class is constructed by the reference's
constructor code -- the real callers of the 21-class `PCONV` surface (extension/main.cpp) --
first over the product shim (ctypes -> libpconv_hip.so; construction needs no GPU), then over
the oracle module, where the reference wrappers' forward runs and is compared, bit for bit,
with this repo's own operator layer on the same inputs.

Runs only where the reference tree exists (the authoring container)."""
import importlib
import os
import sys

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "PCONV_operator")),
                                reason="reference tree absent (GPU box)")


def _to_cpu_instead_of_cuda(original):
    """Tensor.to with every 'cuda:N' target answered by the CPU: the reference's PseudoGDNV2 /
    PseudoQUANTV2 / PseudoDQUANT move their parameters to 'cuda:N' inside __init__
    (PseudoContextV2.py:155,166,247), which a host without a GPU cannot do"""

    def is_cuda(a):
        return (isinstance(a, str) and a.startswith("cuda")) or (isinstance(a, torch.device) and a.type == "cuda")

    def to(self, *args, **kwargs):
        args = tuple("cpu" if is_cuda(a) else a for a in args)
        kwargs = {k: ("cpu" if is_cuda(v) else v) for k, v in kwargs.items()}
        return original(self, *args, **kwargs)

    return to


class _RefOperators(object):
    """imports /root/reference/PCONV_operator with sys.modules['PCONV'] / ['coder'] pointing at
    `ops_module` / `coder_module`; restores sys.modules and sys.path on exit"""

    def __init__(self, ops_module, coder_module, cpu_devices):
        self.ops_module, self.coder_module, self.cpu_devices = ops_module, coder_module, cpu_devices

    def __enter__(self):
        self.saved = {k: v for k, v in sys.modules.items()
                      if k in ("PCONV", "coder") or k == "PCONV_operator" or k.startswith("PCONV_operator.")}
        for k in self.saved:
            del sys.modules[k]
        sys.modules["PCONV"] = self.ops_module
        sys.modules["coder"] = self.coder_module
        sys.path.insert(0, REF)
        try:
            mod = importlib.import_module("PCONV_operator")
        except BaseException:
            self.__exit__(None, None, None)
            raise
        assert os.path.dirname(mod.__file__) == os.path.join(REF, "PCONV_operator")
        if self.cpu_devices:
            self.tensor_to = torch.Tensor.to
            torch.Tensor.to = _to_cpu_instead_of_cuda(self.tensor_to)
        return mod

    def __exit__(self, *exc):
        if getattr(self, "tensor_to", None) is not None:
            torch.Tensor.to, self.tensor_to = self.tensor_to, None
        sys.path.remove(REF)
        for k in [k for k in sys.modules if k in ("PCONV", "coder") or k == "PCONV_operator" or k.startswith("PCONV_operator.")]:
            del sys.modules[k]
        sys.modules.update(self.saved)
        return False


def _construct_all(R):
    """every nn.Module wrapper of the reference package, built the way model_zoo_v2.py /
    pseudo_codec.py build them (model_zoo_v2.py:36-211,214-334; pseudo_codec.py:27-213)"""
    ctx = R.PseudoContextV2(16, True, device=0)
    ctx_plain = R.PseudoContextV2(16, False, rt=20, device=0)
    ectx = R.PseudoEntropyContext(16, 1, True, device=0)
    ectx0 = R.PseudoEntropyContext(16, 0, True, device=0)
    wave = R.EntropyContextNew(16, 18, True, device=0)
    built = {
        "PseudoContextV2": ctx, "PseudoContextV2(opt=False)": ctx_plain, "PseudoEntropyContext": ectx,
        "PseudoEntropyContext(v0)": ectx0, "EntropyContextNew": wave,
        "SphereSlice": R.SphereSlice(16, pad=0, opt=True, device=0),
        "SphereUslice": R.SphereUslice(16, pad=0, opt=True, device=0),
        "SphereUslice(pad=2)": R.SphereUslice(16, pad=2, opt=True, device=0),
        "PseudoPadV2": R.PseudoPadV2(1, 16, ctx, device=0),
        "PseudoFillV2": R.PseudoFillV2(0, 16, ctx, device=0),
        "PseudoFillV2(trim)": R.PseudoFillV2(2, 16, ctx, fvalue=0, trim=1, device=0),
        "PseudoGDNV2": R.PseudoGDNV2(192, 16, ctx, 0),
        "PseudoGDNV2(inverse)": R.PseudoGDNV2(192, 16, ctx, 0, inverse=True),
        "PseudoEntropyPad": R.PseudoEntropyPad(2, 16, ectx, device=0),
        "PseudoQUANTV2": R.PseudoQUANTV2(192, 8, 16, ctx, top_alpha=0.0001, device_id=0, ntop=2),
        "PseudoDQUANT": R.PseudoDQUANT(192, 8, 16, ctx, device_id=0),
        "Dtow": R.Dtow(2, True, 0), "Dtow(w2d)": R.Dtow(2, False, 0),
        "ContextReshape": R.ContextReshape(14, 0),
        "EntropyGmm": R.EntropyGmm(3, 0, 0),
        "EntropyGmmTable": R.EntropyGmmTable(8, 3.5, 3, 65536, device=0),
        "EntropyBatchGmmTable": R.EntropyBatchGmmTable(8, 3.5, 3, 65536, device=0),
        "MaskConv2": R.MaskConv2(14, 1, 3, 5, False, 0),
        "MaskConv2(hidden)": R.MaskConv2(14, 3, 3, 5, True, 0),
        "EntropyConv2": R.EntropyConv2(16, 14, 1, 3, 5, wave, 2, 2, False, True, 0),
        "EntropyConv2Batch": R.EntropyConv2Batch(16, 14, 3, 3, 5, wave, 2, 2, 3, True, True, 0),
        "EntropyConv2Batch(last)": R.EntropyConv2Batch(16, 14, 3, 3, 5, wave, 2, 0, 3, True, False, 0),
        "EntropyCtxPadRun2": R.EntropyCtxPadRun2(2, 16, 14, wave, False, 0),
        "EntropyCtxPadRun2(input)": R.EntropyCtxPadRun2(2, 16, 14, wave, True, 0),
        "EntropyAdd": R.EntropyAdd(16, 42, 14, 2, wave, 0),
        "DExtract2": R.DExtract2(16, 14, True, wave, 0),
        "DExtract2Batch": R.DExtract2Batch(16, 126, wave, 0),
        "DInput2": R.DInput2(14, 16, wave, 2, -3.5, 3, 0),
        "MultiProject": R.MultiProject(171, 256, 0.6, False, 0),
        "MultiProjectM": R.MultiProjectM(64, 96, [0.0, 1.0], [0.0, 0.5], 0.6, False, 0),
        "SSIM": R.SSIM(11, 3), "StubMask": R.StubMask(56), "Extract": R.Extract(56), "DropGrad": R.DropGrad(True),
    }
    return built


def test_reference_wrappers_construct_on_the_product_shim():
    from pseudocylindrical_convolution_amd import PCONV as shim, coder as coder_shim
    with _RefOperators(shim, coder_shim, not torch.cuda.is_available()) as R:
        built = _construct_all(R)
        exported = [n for n in dir(R) if isinstance(getattr(R, n), type) and issubclass(getattr(R, n), torch.nn.Module)]
        for name in exported:                                      # nothing of the package left unconstructed
            assert any(type(m).__name__ == name for m in built.values()), name
        # each wrapper holds an op of the product shim, keyed by GPU id (BaseOpModule.py:9)
        for key, m in built.items():
            op = getattr(m, "op", None)
            if isinstance(op, dict):
                assert list(op) == [0], key
                assert type(op[0]).__module__ == shim.__name__, key


def _inputs():
    g = torch.Generator().manual_seed(11)
    return {
        "img": torch.rand(1, 3, 256, 512, generator=g),
        "tiles": torch.rand(16, 3, 16, 512, generator=g),
        "feat": torch.rand(16, 8, 4, 128, generator=g) - 0.5,
        "code": torch.rand(16, 8, 1, 32, generator=g),
    }


class _OnCpu(object):
    """the reference picks a wrapper's op by `x.device.index` (SphereSlice.py:10-13), which is None
    for a CPU tensor: the op built for GPU id 0 also answers for None"""

    def __init__(self, package):
        self.package = package

    def __getattr__(self, name):
        cls = getattr(self.package, name)

        def build(*args, **kwargs):
            m = cls(*args, **kwargs)
            for sub in m.modules():
                op = getattr(sub, "op", None)
                if isinstance(op, dict) and 0 in op:
                    op.setdefault(None, op[0])
            return m

        return build


def _run_chain(P):
    """a forward chain through the wrappers that carry arithmetic, on whatever backend is active"""
    x = _inputs()
    P = _OnCpu(P)
    ctx = P.PseudoContextV2(16, True, device=0)
    out = {}
    with torch.no_grad():
        t = P.SphereSlice(16, pad=0, opt=True, device=0)(x["img"])
        out["slice"] = t.clone()
        out["pad"] = P.PseudoPadV2(2, 16, ctx, device=0)(t).clone()
        out["uslice"] = P.SphereUslice(16, pad=2, opt=True, device=0)(out["pad"]).clone()
        out["fill"] = P.PseudoFillV2(0, 16, ctx, device=0)(x["feat"].clone()).clone()
        torch.manual_seed(3)
        gdn = P.PseudoGDNV2(8, 16, ctx, 0)
        out["gdn"] = gdn(x["feat"].clone()).clone()
        torch.manual_seed(3)
        igdn = P.PseudoGDNV2(8, 16, ctx, 0, inverse=True)
        out["igdn"] = igdn(x["feat"].clone()).clone()
        out["dtow"] = P.Dtow(2, True, 0)(x["feat"]).clone()
        out["wtod"] = P.Dtow(2, False, 0)(x["feat"]).clone()
        torch.manual_seed(4)
        q = P.PseudoQUANTV2(8, 8, 16, ctx, top_alpha=0.0001, device_id=0, ntop=2).eval()
        val, idx = q(x["code"])
        out["quant_val"], out["quant_idx"] = val.clone(), idx.clone()
        torch.manual_seed(4)
        dq = P.PseudoDQUANT(8, 8, 16, ctx, device_id=0)
        out["dquant"] = dq(idx).clone()
        mp = P.MultiProject(43, 64, 0.6, False, 0)
        out["project"] = mp(x["img"]).clone()
    return out


def test_reference_wrappers_forward_equals_this_repos_operator_layer(oracle_backend):
    """same native module underneath (the CPU oracle), two operator layers on top: the
    reference's own Python and this repo's.  Pins the wrappers' call-site semantics
    (argument order, buffer handling, the GDN formula of PseudoContextV2.py:133-216,
    quantiser bridge :218-255) to the reference's code."""
    from oracle import pconv_cpu, coder_cpu
    from pseudocylindrical_convolution_amd import PCONV_operator as mine
    with _RefOperators(pconv_cpu, coder_cpu, True) as R:
        ref_out = _run_chain(R)
    my_out = _run_chain(mine)
    assert sorted(ref_out) == sorted(my_out)
    for k in ref_out:
        assert ref_out[k].shape == my_out[k].shape, k
        assert torch.equal(ref_out[k], my_out[k]), "%s: max abs diff %g" % (k, (ref_out[k] - my_out[k]).abs().max())
