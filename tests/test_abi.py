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


def test_lds_dma_offsets_beyond_32_bits_are_refused():
    """The LDS-DMA paths of both convolution kernels address a chunk of input channels as a 64-bit
    uniform base + a 32-bit byte offset per lane (csrc/conv.hip, csrc/wino.hip).  A view whose chunk
    spans 4 GiB or more must be refused on the host, before any launch (this is the regime of
    pseudo_pad.cu:106's int32 count: SURVEY 7.3) -- a wrapped offset would read another tile's data."""
    import ctypes
    from pseudocylindrical_convolution_amd import _native
    lib = _native.hip_lib()
    dummy = 4096  # never dereferenced: the checks come first
    cin, h, w, cout = 192, 6, 66, 192
    ok_cs = h * w

    def views(cs, n):
        rows = [[cin * cs, cs, w], [cout * (h - 2) * (w - 2), (h - 2) * (w - 2), w - 2]] + [[0, 0, 0]] * (n - 2)
        flat = [v for r in rows for v in r]
        return (ctypes.c_longlong * len(flat))(*flat)

    # Winograd: 4 channels per chunk -> 3 * cs * 4 bytes must stay below 2^32
    big = (1 << 32) // 12 + 1
    v = views(big, 3)
    rc = lib.pconv_conv3x3_wino(dummy, dummy, dummy, dummy, 1, cin, h, w, cout, 0, None, None, 0, None, 0, 0,
                                ctypes.addressof(v), None)
    assert rc < 0 and b"32-bit byte offsets" in lib.pconv_last_error()
    # direct kernel: 16 channels per chunk
    big = (1 << 32) // 60 + 1
    v = views(big, 4)
    rc = lib.pconv_conv2d(dummy, dummy, dummy, dummy, 1, cin, h, w, cout, 3, 1, 0, None, None, 0, None, None, 0, 0,
                          ctypes.addressof(v), None)
    assert rc < 0 and b"32-bit byte offsets" in lib.pconv_last_error()
    # ... and the output / residual side of the quad ways out (their strides are independent of the input's)
    big = (1 << 32) // (31 * 4) + 1
    rows = [[cin * ok_cs, ok_cs, w], [cout * big, big, w - 2], [0, 0, 0], [0, 0, 0]]
    flat = [x for r in rows for x in r]
    v = (ctypes.c_longlong * len(flat))(*flat)
    rc = lib.pconv_conv2d(dummy, dummy, dummy, dummy, 1, cin, h, w, cout, 3, 1, 0, None, None, 0, None, None, 0, 0,
                          ctypes.addressof(v), None)
    assert rc < 0 and b"output / residual channel stride" in lib.pconv_last_error()
    # the largest stride of the codec (8 frames at 2048x4096: a 1/2-scale 192-channel tile-batch tensor,
    # channel stride 68 * 2052 floats) is nowhere near the limit
    assert (15 * 68 * 2052 + 67 * 2052 + 2052) * 4 < (1 << 32)
