"""The drop-in surface: module / class / method names of the reference's `PCONV`,
`coder` and `PCONV_operator`, and -- when the reference tree is present -- its own
model_zoo_v2.py importing on top of this package unchanged."""
import importlib.util
import inspect
import os
import sys

import pytest

REF = "/root/reference"


def test_pconv_module_has_the_21_classes_with_reference_methods():
    from pseudocylindrical_convolution_amd import PCONV
    expect = {
        "ProjectsOp": ["to", "forward", "backward"], "DtowOp": ["to", "forward", "backward"],
        "ContextReshapeOp": ["to", "forward", "backward"], "EntropyGmmOp": ["to", "forward", "backward"],
        "MaskConstrainOp": ["to", "forward", "backward"], "SphereSliceOp": ["to", "forward", "backward"],
        "SphereUsliceOp": ["to", "forward", "backward"], "EntropyGmmTableOp": ["to", "forward", "forward_batch"],
        "EntropyContextOp": ["to", "start_context", "addr"],
        "EntropyCtxPadRun2Op": ["to", "restart", "forward", "backward"],
        "DExtract2Op": ["to", "restart", "forward", "forward_batch"], "DInput2Op": ["to", "restart", "forward"],
        "EntropyConv2Op": ["to", "restart", "forward", "forward_act", "forward_batch", "forward_act_batch"],
        "PseudoContextOp": ["to", "start_context", "addr", "produce_fill_param"],
        "PseudoPadOp": ["to", "forward", "backward"], "PseudoFillOp": ["to", "forward", "backward"],
        "PseudoEntropyContextOp": ["to", "start_context", "addr"], "PseudoEntropyPadOp": ["to", "forward", "backward"],
        "PseudoQuantOp": ["to", "forward", "backward"], "PseudoDQuantOp": ["to", "forward"],
        "EntropyAddOp": ["to", "restart", "forward"],
    }
    assert sorted(PCONV.__all__) == sorted(expect)
    for cls, methods in expect.items():
        for m in methods:
            assert callable(getattr(getattr(PCONV, cls), m)), "%s.%s" % (cls, m)
    # constructor arity as bound in extension/main.cpp
    arity = {"ProjectsOp": 8, "DtowOp": 4, "ContextReshapeOp": 3, "EntropyGmmOp": 4, "MaskConstrainOp": 4,
             "SphereSliceOp": 6, "SphereUsliceOp": 6, "EntropyGmmTableOp": 7, "EntropyContextOp": 5,
             "EntropyCtxPadRun2Op": 7, "DExtract2Op": 6, "DInput2Op": 8, "EntropyConv2Op": 11, "PseudoContextOp": 5,
             "PseudoPadOp": 5, "PseudoFillOp": 8, "PseudoEntropyContextOp": 6, "PseudoEntropyPadOp": 5,
             "PseudoQuantOp": 10, "PseudoDQuantOp": 6, "EntropyAddOp": 7}
    for cls, n in arity.items():
        params = [p for p in inspect.signature(getattr(PCONV, cls).__init__).parameters if p != "self"]
        assert len(params) == n, "%s takes %d constructor arguments, reference binds %d" % (cls, len(params), n)


def test_operator_package_exports_the_reference_names():
    from pseudocylindrical_convolution_amd import PCONV_operator as P
    for name in ["MultiProject", "MultiProjectM", "Dtow", "SSIM", "ModuleSaver", "Logger", "EntropyGmm",
                 "ContextReshape", "DropGrad", "MaskConv2", "SphereSlice", "SphereUslice", "StubMask", "Extract",
                 "EntropyGmmTable", "EntropyBatchGmmTable", "EntropyContextNew", "EntropyConv2", "EntropyConv2Batch",
                 "EntropyCtxPadRun2", "DExtract2", "DInput2", "DExtract2Batch", "EntropyAdd", "PseudoFillV2",
                 "PseudoContextV2", "PseudoGDNV2", "PseudoPadV2", "PseudoEntropyContext", "PseudoEntropyPad",
                 "PseudoQUANTV2", "PseudoDQUANT"]:
        assert hasattr(P, name), name


def test_coder_module_surface():
    from pseudocylindrical_convolution_amd import coder
    for m in ["encode", "decode", "encodes", "decodes", "start_encoder", "end_encoder", "start_decoder"]:
        assert callable(getattr(coder.coder, m))


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree absent (GPU box)")
def test_reference_model_zoo_imports_unchanged_on_top_of_the_dropin():
    import pseudocylindrical_convolution_amd as pkg
    pkg.install_dropin()
    spec = importlib.util.spec_from_file_location("ref_model_zoo_v2", os.path.join(REF, "model_zoo_v2.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                                   # its `from PCONV_operator import ...` resolve here
    for name in ["EncoderV2", "DecoderV2", "ClipData", "ResidualBlockDown", "AttentionBlock"]:
        assert hasattr(mod, name)
    for k in ("PCONV", "coder", "PCONV_operator"):
        assert sys.modules[k].__name__.startswith("pseudocylindrical_convolution_amd")


def test_module_retargets_its_op_when_moved():
    """BaseOpModule keeps one native op per GPU id and re-keys it on .to(device)"""
    import torch
    from pseudocylindrical_convolution_amd.PCONV_operator import BaseOpModule

    class Fake(object):
        def __init__(self):
            self.dev = 0

        def to(self, d):
            self.dev = d

    m = BaseOpModule(0)
    m.op = {0: Fake()}
    fn = lambda t: t  # dtype-only conversion: nothing moves
    m._apply(fn)
    assert list(m.op) == [0]
    m.custom_op_to(torch.device("cuda", 3))
    assert list(m.op) == [3] and m.op[3].dev == 3
