"""`coder` -- the arithmetic-coder module of the reference, over libpconv_coder.so.

Mirrors the pybind class `coder.coder` (coder/python.cpp:63-73): same method
names, argument meaning and tensor conventions.  Where the reference throws a
C string (surfaced by pybind as RuntimeError) this raises CoderError, a
RuntimeError subclass.
"""
import ctypes

import torch

from ._native import coder_lib

__all__ = ["coder", "CoderError"]


class CoderError(RuntimeError):
    pass


def _i32_cpu(t, what):
    if t.device.type != "cpu" or t.dtype != torch.int32:
        # the reference reads data_ptr<int>() of whatever it is given (python.cpp:23-24)
        raise CoderError("%s must be an int32 CPU tensor, got %s on %s" % (what, t.dtype, t.device))
    return t if t.is_contiguous() else t.contiguous()


class coder(object):
    """coder(filename): 32-bit arithmetic coder bound to one code file
    (coder/coder.h:8-59)."""

    def __init__(self, filename):
        self._lib = coder_lib()
        self._h = self._lib.pconv_coder_new(str(filename).encode())
        if not self._h:
            raise CoderError("cannot allocate coder")
        self.filename = str(filename)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.pconv_coder_free(h)

    def _check(self, rc):
        if rc < 0:
            raise CoderError((self._lib.pconv_coder_error(self._h) or b"").decode() + " (%d)" % rc)
        return rc

    # -- encoder ----------------------------------------------------------
    def start_encoder(self):
        self._check(self._lib.pconv_coder_start_encoder(self._h))

    def encode(self, table, ncode, total, symbol):
        """one symbol; `table` holds ncode+1 cumulative counts (python.cpp:4-12)"""
        t = table.to("cpu").to(torch.int32).contiguous()
        self._check(self._lib.pconv_coder_encode(self._h, t.data_ptr(), int(ncode), int(total), int(symbol)))

    def encodes(self, table, ncode, symbols, num):
        """`num` symbols; table int32 [n, ncode+1], symbols int32 [n] (python.cpp:22-40)"""
        t = _i32_cpu(table, "table")
        s = _i32_cpu(symbols, "symbols")
        num = int(num)
        if num * (int(ncode) + 1) > t.numel() or num > s.numel():
            raise CoderError("encodes: %d symbols do not fit the given tensors" % num)
        self._check(self._lib.pconv_coder_encodes(self._h, t.data_ptr(), int(ncode), s.data_ptr(), num))

    def end_encoder(self):
        self._check(self._lib.pconv_coder_end_encoder(self._h))

    def bytes(self):
        """encoded bytes of the last finished stream (not in the reference API)"""
        n = ctypes.c_size_t(0)
        p = self._lib.pconv_coder_bytes(self._h, ctypes.byref(n))
        return ctypes.string_at(p, n.value) if n.value else b""

    # -- decoder ----------------------------------------------------------
    def start_decoder(self):
        self._check(self._lib.pconv_coder_start_decoder(self._h))

    def decode(self, table, ncode, total):
        t = table.to("cpu").to(torch.int32).contiguous()
        return self._check(self._lib.pconv_coder_decode(self._h, t.data_ptr(), int(ncode), int(total)))

    def decodes(self, table, ncode, num):
        """-> float32 CPU tensor of table.size(0) entries, the first `num` decoded
        (python.cpp:41-61)"""
        t = _i32_cpu(table, "table")
        num = int(num)
        if num * (int(ncode) + 1) > t.numel():
            raise CoderError("decodes: %d rows requested, table has fewer" % num)
        out = torch.zeros(t.shape[0], dtype=torch.float32)
        self._check(self._lib.pconv_coder_decodes(self._h, t.data_ptr(), int(ncode), out.data_ptr(), num))
        return out
