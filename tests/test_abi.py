"""The C-ABI libraries load and export every symbol the public headers declare
(no kernel is launched: this runs without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pconv_\w+)\s*\(", src)))


def test_hip_library_exports_every_declared_symbol():
    from pseudocylindrical_convolution_amd import _native
    lib = _native.hip_lib()
    names = declared("pconv_hip.h")
    assert len(names) > 40
    for name in names:
        assert hasattr(lib, name), "libpconv_hip.so lacks %s" % name
    # the ctypes table covers the same set
    assert set(_native.declared_hip_symbols()) == set(names)
    assert lib.pconv_abi_version() == 1


def test_coder_library_exports_every_declared_symbol():
    from pseudocylindrical_convolution_amd import _native
    lib = _native.coder_lib()
    names = declared("pconv_coder.h")
    for name in names:
        assert hasattr(lib, name), "libpconv_coder.so lacks %s" % name
    assert set(_native.declared_coder_symbols()) == set(names)


def test_argument_counts_match_the_header():
    from pseudocylindrical_convolution_amd import _native
    src = open(os.path.join(ROOT, "include", "pconv_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = dict(re.findall(r"\b(pconv_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S))
    for name, sig in _native._HIP_SIGNATURES.items():
        args = protos[name].strip()
        n = 0 if args in ("", "void") else len(args.split(","))
        assert n == len(sig), "%s: header has %d arguments, binding %d" % (name, n, len(sig))


def test_bad_arguments_are_reported_not_launched():
    from pseudocylindrical_convolution_amd import _native
    lib = _native.hip_lib()
    # null pointers are rejected on the host before any launch
    rc = lib.pconv_dtow(None, None, 1, 4, 2, 2, 2, 1, None)
    assert rc < 0
    assert b"dtow" in lib.pconv_last_error()
    with pytest.raises(_native.PconvError):
        _native.call("pconv_host_tile_widths", None, 16, 512, 1024, None)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from pseudocylindrical_convolution_amd import _native
    monkeypatch.setattr(_native, "HERE", str(tmp_path))
    monkeypatch.setattr(_native, "_hip", None)
    with pytest.raises(_native.PconvError, match="no CPU fallback"):
        _native.hip_lib()


def test_ops_refuse_cpu_tensors():
    import torch
    from pseudocylindrical_convolution_amd import PCONV
    from pseudocylindrical_convolution_amd._native import PconvError
    with pytest.raises(PconvError, match="GPU tensor"):
        PCONV.DtowOp(2, True, 0, False).forward(torch.zeros(1, 4, 2, 2))
