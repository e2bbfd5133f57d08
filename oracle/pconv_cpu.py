"""TEST INFRASTRUCTURE -- CPU oracle with the surface of the reference's `PCONV`.

A second, independent implementation of the 21 native op classes
(/root/reference/extension/main.cpp:4-137) on CPU tensors, over the C restatement
in oracle/pconv_oracle.c.  It follows the reference's C++ op classes (reshape /
counters / buffer reuse), not the product's shim.  Only tests/, smoke() and
bench.py's cpu_baseline leg may import it; they plug it under the operator layer
with `PCONV_operator.backend.use(oracle.pconv_cpu, oracle.coder_cpu)`.

Parity status: "parity unpinned" by reference data for these kernels (the
reference has no vectors and its CUDA build cannot run here); pinned by the
invariants of tests/test_oracle_properties.py.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
DEVICE_FMT = "cpu"
_lib = None


def build():
    """compile oracle/pconv_oracle.c (and oracle/_ref when the reference tree is present)"""
    subprocess.run(["make", "-s", "-C", HERE], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "_build", "libpconv_oracle.so")
        if not os.path.exists(path):
            build()
        _lib = ctypes.CDLL(path)
        _lib.orc_cal_npart_hw_v2.restype = ctypes.c_int
        _lib.orc_entropy_context.restype = ctypes.c_int
    return _lib


def host_cpu_share():
    """CPUs this process may really use: the affinity mask, cut down to the cgroup's CPU quota
    when there is one (a GPU box shows all 256 host CPUs to a container that owns 16 of them;
    256 OpenMP threads on a 16-CPU share crawl)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                fields = f.read().split()
            if path.endswith("cpu.max"):
                if fields[0] != "max":
                    n = min(n, max(1, int(int(fields[0]) / int(fields[1]))))
            else:
                quota = int(fields[0])
                if quota > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, int(quota / int(f.read()))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def set_num_threads(n=None):
    """threads of the oracle's OpenMP loops (default: the host CPU share, at most 16 -- what a one-GPU
    box gives its job); returns the count"""
    n = int(n) if n else max(1, min(host_cpu_share(), 16))
    lib().orc_set_num_threads(ctypes.c_int(n))
    return n


def set_detmath(on):
    """True: CDF / quantiser tables use the product's published erf/exp polynomials
    (bit-exact comparisons); False: libm, as a stand-in for the reference's CUDA math."""
    lib().orc_set_detmath(ctypes.c_int(1 if on else 0))


# entropy-conv summation order: 0 = the reference's (128-thread tree), 1 = the product's (tap-major, 64 lanes:
# what the engine, the per-op kernel and every stream use), 2 = the causal-compact order of the round-4
# experiment (only the unmasked entries, by kh + kw, kh, channel: tools/experiments/)
CONV_ORDER = 1


def _p(t):
    if t is None:
        return None
    if isinstance(t, np.ndarray):
        return ctypes.c_void_p(t.ctypes.data)
    assert t.device.type == "cpu" and t.is_contiguous(), "oracle works on contiguous CPU tensors"
    return ctypes.c_void_p(t.data_ptr())


I = ctypes.c_int
F = ctypes.c_float
L = ctypes.c_longlong
_contexts = {}


def _ctx(addr):
    return _contexts[addr]


class _Base(object):

    def __init__(self, device=0, timeit=False):
        self.device_ = device
        self.top = {}
        self.shape_ = None

    def to(self, device):
        self.device_ = device

    def _reshape(self, *shape):
        if self.shape_ == tuple(shape):
            return False
        self.shape_ = tuple(shape)
        return True

    def _top(self, slot, shape, zero=False):
        t = self.top.get(slot)
        if t is None or tuple(t.shape) != tuple(shape):
            t = torch.zeros(shape, dtype=torch.float32) if zero else torch.empty(shape, dtype=torch.float32)
            self.top[slot] = t
        return t


def widths_v3(weight, npart, height, width):
    out = np.zeros(npart, np.int32)
    lib().orc_cal_npart_hw_v3(I(height), I(width), I(npart), _p(np.asarray(weight, np.float32)), _p(out))
    return out


# -- contexts -------------------------------------------------------------------
class _Context(_Base):

    def __init__(self, npart, rt, weight, device=0, timeit=False):
        super().__init__(device, timeit)
        self.npart_, self.rt_ = npart, rt
        self.weight_ = np.asarray(list(weight), np.float32)
        self.data_width_ = -1
        self.cache = {}
        self.addr_ = "oracle-ctx-%d" % id(self)
        _contexts[self.addr_] = self

    def addr(self):
        return self.addr_

    def start_context(self, width):
        if width != self.data_width_:
            self.cache = {}
        self.data_width_ = width

    def hindex(self, height, width):
        key = ("hindex", width)
        if key not in self.cache:
            self.cache[key] = widths_v3(self.weight_, self.npart_, height * self.npart_, width)
        return self.cache[key]

    def produce_fill_param(self, height, width):
        return torch.from_numpy(self.hindex(height, width).copy())


class PseudoContextOp(_Context):

    def produce_param(self, channel, height, width, pad):
        key = ("param", width, channel, pad, height)
        if key not in self.cache:
            hidx = self.hindex(height, width)
            n = self.npart_ * 2 * max(pad, 1) * width
            dst, src = np.zeros(n, np.int64), np.zeros(n, np.int64)
            pcol, pt = np.zeros(n, np.int32), np.zeros(n, np.float32)
            h2 = np.zeros(self.npart_ * 2 * max(pad, 1), np.int32)
            if pad > 0:
                lib().orc_pseudo_context(_p(hidx), _p(h2), _p(dst), _p(src), _p(pcol), _p(pt), I(channel),
                                         I(height), I(width), I(self.npart_), I(pad))
            self.cache[key] = (hidx, h2, dst, src, pcol, pt)
        return self.cache[key]


class PseudoEntropyContextOp(_Context):

    def __init__(self, npart, rt, context_version, weight, device=0, timeit=False):
        super().__init__(npart, rt, weight, device, timeit)
        self.context_version_ = context_version

    def produce_param(self, channel, height, width, pad):
        key = ("param", width, channel, pad, height)
        if key not in self.cache:
            hidx = self.hindex(height, width)
            n = self.npart_ * 2 * pad * width
            dst, src = np.zeros(n, np.int64), np.zeros(n, np.int64)
            pcol, pt = np.zeros(n, np.int32), np.zeros(n, np.float32)
            h2 = np.zeros(self.npart_ * 2 * pad, np.int32)
            lib().orc_pseudo_entropy_context(_p(hidx), _p(h2), _p(dst), _p(src), _p(pcol), _p(pt), I(channel),
                                             I(height), I(width), I(self.npart_), I(pad), I(self.context_version_))
            self.cache[key] = (hidx, h2, dst, src, pcol, pt)
        return self.cache[key]


class EntropyContextOp(_Context):

    def produce_param_group(self, height, width):
        key = ("group", width, height)
        if key not in self.cache:
            hidx = self.hindex(height, width)
            rows = height * self.npart_
            idx = np.zeros(rows * width, np.int32)
            start = np.zeros(rows + width, np.int32)
            lib().orc_wavefront(_p(hidx), I(self.npart_), I(height), I(width), _p(idx), _p(start))
            self.cache[key] = (idx, start)
        return self.cache[key]

    def produce_param(self, channel, height, width, pad):
        key = ("param", width, channel, pad, height)
        if key not in self.cache:
            hidx = self.hindex(height, width)
            n = self.npart_ * 2 * pad * width
            dst, src = np.zeros(n, np.int64), np.zeros(n, np.int64)
            pcol, pt = np.zeros(n, np.int32), np.zeros(n, np.float32)
            h2 = np.zeros(self.npart_ * 2 * pad, np.int32)
            cap = n + self.npart_ * (height + 2 * pad) * pad
            lst = np.zeros(cap * 3, np.int64)
            pad_idx = np.zeros(height * self.npart_ + width + 2 * pad + 1, np.int32)
            total = lib().orc_entropy_context(_p(hidx), _p(h2), _p(dst), _p(src), _p(pcol), _p(pt), _p(lst),
                                              I(cap), _p(pad_idx), I(channel), I(height), I(width),
                                              I(self.npart_), I(pad))
            assert total >= 0
            self.cache[key] = (hidx, h2, dst, src, pcol, pt, lst, pad_idx)
        return self.cache[key]


# -- transform ops -----------------------------------------------------------------
class DtowOp(_Base):

    def __init__(self, stride, d2w, device=0, timeit=False):
        super().__init__(device, timeit)
        self.stride_, self.d2w_ = stride, d2w

    def forward(self, x):
        n, c, h, w = x.shape
        s = self.stride_
        shape = (n, c // (s * s), h * s, w * s) if self.d2w_ else (n, c * s * s, h // s, w // s)
        out = self._top(0, shape)
        lib().orc_dtow(_p(x), _p(out), I(n), I(c), I(h), I(w), I(s), I(1 if self.d2w_ else 0))
        return [out]

    def backward(self, grad):
        # dtow_cuda.cu:105-167: bottom_diff[index] = top_diff[where the forward sent index]
        n, c, h, w = grad.shape
        s = self.stride_
        shape = (n, c * s * s, h // s, w // s) if self.d2w_ else (n, c // (s * s), h * s, w * s)
        out = self._top(1, shape)
        lib().orc_dtow(_p(grad.contiguous()), _p(out), I(n), I(c), I(h), I(w), I(s), I(0 if self.d2w_ else 1))
        return [out]


class ContextReshapeOp(_Base):

    def __init__(self, ngroup, device=0, timeit=False):
        super().__init__(device, timeit)
        self.ngroup_ = ngroup

    def forward(self, x):
        n, c, h, w = x.shape
        cpg = c // self.ngroup_
        self.shape_ = (n, c, h, w)
        out = self._top(0, (n * h * w * self.ngroup_, cpg))
        lib().orc_context_reshape(_p(x), _p(out), I(n), I(c), I(h), I(w), I(cpg))
        return [out]

    def backward(self, grad):
        n, c, h, w = self.shape_
        out = self._top(1, (n, c, h, w))
        lib().orc_context_reshape_backward(_p(out), _p(grad.contiguous()), I(n), I(c), I(h), I(w), I(c // self.ngroup_))
        return [out]


class EntropyGmmOp(_Base):

    def __init__(self, num_gaussian, ignore_label, device=0, timeit=False):
        super().__init__(device, timeit)
        self.num_gaussian_ = num_gaussian

    def forward(self, weight, delta, mean, label):
        m, ng = weight.shape
        loss = self._top(0, (m,))
        d = [self._top(i + 1, (m, ng)) for i in range(3)] + [self._top(4, (m, 1))]
        lib().orc_gmm_loss(_p(weight), _p(delta), _p(mean), _p(label), _p(d[0]), _p(d[1]), _p(d[2]), _p(d[3]),
                           _p(loss), I(m), I(ng))
        return [loss]

    def backward(self, grad):
        # entropy_gmm_cuda.cu:95-127: the stored per-row derivatives times the incoming gradient
        g = grad.reshape(-1, 1)
        return [self.top[1] * g, self.top[2] * g, self.top[3] * g, self.top[4] * g]


class MaskConstrainOp(_Base):

    def __init__(self, constrain, ngroup, device=0, timeit=False):
        super().__init__(device, timeit)
        self.constrain_, self.ngroup_ = constrain, ngroup

    def forward(self, w):
        nout, cin, k, _ = w.shape
        lib().orc_mask_constrain(_p(w), I(nout), I(cin), I(k), I(self.ngroup_), I(self.constrain_))

    def backward(self, grad):
        self.forward(grad)


class SphereSliceOp(_Base):

    def __init__(self, npart, interp_type, pad, weight, device=0, timeit=False):
        super().__init__(device, timeit)
        self.npart_, self.pad_ = npart, pad
        self.weight_ = np.asarray(list(weight), np.float32)
        self.tabs = {}

    def forward(self, x):
        n, c, h, w = x.shape
        key = (h, w)
        if key not in self.tabs:
            tidx = np.zeros(2 * self.npart_, np.int32)
            hinv = np.zeros(2 * h, np.int32)
            h_out = lib().orc_cal_npart_hw_v2(I(h), I(w), I(self.npart_), _p(self.weight_), _p(tidx), _p(hinv))
            assert h_out >= 0, "height must be a multiple of npart"
            param = np.zeros(self.npart_ * w * 5, np.float32)
            lib().orc_slice_param(I(self.npart_), I(w), _p(tidx), _p(param))
            self.tabs[key] = (tidx, param, h_out)
        tidx, param, h_out = self.tabs[key]
        p = self.pad_
        out = self._top(0, (n * self.npart_, c, h_out + 2 * p, w + 2 * p), zero=True)
        lib().orc_slice_forward(_p(x), _p(out), _p(param), _p(tidx), I(n * self.npart_), I(c), I(h_out), I(w),
                                I(h), I(self.npart_), I(p))
        self.shape_ = (n, c, h, w)
        return [out]

    def backward(self, grad):
        n, c, h, w = self.shape_
        tidx, param, h_out = self.tabs[(h, w)]
        gin = self._top(1, (n, c, h, w))
        lib().orc_slice_backward(_p(gin), _p(grad.contiguous()), _p(param), _p(tidx), I(n * self.npart_), I(c), I(h_out),
                                 I(w), I(h), I(self.npart_), I(self.pad_))
        return [gin]


class SphereUsliceOp(_Base):

    def __init__(self, npart, interp_type, pad, weight, device=0, timeit=False):
        super().__init__(device, timeit)
        self.npart_, self.pad_ = npart, pad
        self.weight_ = np.asarray(list(weight), np.float32)
        self.tabs = {}

    def forward(self, x):
        p = self.pad_
        tn, c, hp, wp = x.shape
        h, w = hp - 2 * p, wp - 2 * p
        key = (h, w)
        if key not in self.tabs:
            hidx = widths_v3(self.weight_, self.npart_, h * self.npart_, w)
            param = np.zeros(self.npart_ * w * 5, np.float32)
            lib().orc_uslice_param(I(self.npart_), I(w), _p(hidx), _p(param))
            self.tabs[key] = (hidx, param)
        hidx, param = self.tabs[key]
        n_out = tn // self.npart_
        out = self._top(0, (n_out, c, h * self.npart_, w))
        lib().orc_uslice_forward(_p(x), _p(out), _p(param), _p(hidx), I(n_out), I(c), I(h), I(w), I(self.npart_), I(p))
        return [out]

    def backward(self, grad):
        n_out, c, hh, w = grad.shape
        h, p = hh // self.npart_, self.pad_
        hidx, param = self.tabs[(h, w)]
        gin = self._top(1, (n_out * self.npart_, c, h + 2 * p, w + 2 * p))
        lib().orc_uslice_backward(_p(gin), _p(grad.contiguous()), _p(param), _p(hidx), I(n_out), I(c), I(h), I(w),
                                  I(self.npart_), I(p))
        return [gin]


class PseudoPadOp(_Base):

    def __init__(self, pad, npart, ctx_addr, device=0, timeit=False):
        super().__init__(device, timeit)
        self.pad_, self.npart_, self.ctx_ = pad, npart, _ctx(ctx_addr)

    def forward(self, x):
        num, c, h, w = x.shape
        p = self.pad_
        hidx, h2, dst, src, pcol, pt = self.ctx_.produce_param(c, h, w, p)
        out = self._top(0, (num, c, h + 2 * p, w + 2 * p))
        if p == 0:
            out.copy_(x)
            return [out]
        lib().orc_pseudo_pad(_p(x), _p(out), _p(hidx), _p(h2), _p(dst), _p(src), _p(pcol), _p(pt), I(num), I(c),
                             I(h), I(w), I(self.npart_), I(p))
        return [out]

    def backward(self, grad):
        num, c, hp, wp = grad.shape
        p = self.pad_
        h, w = hp - 2 * p, wp - 2 * p
        hidx, h2, dst, src, pcol, pt = self.ctx_.produce_param(c, h, w, p)
        gin = self._top(1, (num, c, h, w))
        lib().orc_pseudo_pad_backward(_p(grad.contiguous()), _p(gin), _p(hidx), _p(h2), _p(dst), _p(src), _p(pcol),
                                      _p(pt), I(num), I(c), I(h), I(w), I(self.npart_), I(p))
        return [gin]


class PseudoFillOp(_Base):

    def __init__(self, pad, npart, fvalue, trim, addr, context_version, device=0, timeit=False):
        super().__init__(device, timeit)
        self.pad_, self.npart_, self.fvalue_, self.trim_ = pad, npart, fvalue, trim
        self.ctx_ = _ctx(addr)

    def forward(self, x):
        num, c, h, w = x.shape
        hidx = self.ctx_.hindex(h, w)
        lib().orc_pseudo_fill(_p(x), _p(hidx), I(num), I(c), I(h), I(w), I(self.npart_), I(self.pad_),
                              I(self.trim_), F(self.fvalue_))
        return [x]

    def backward(self, grad):
        # pseudo_fill_cuda.cu:63-77: the same kernel on the gradient, filling zeros
        num, c, h, w = grad.shape
        hidx = self.ctx_.hindex(h, w)
        lib().orc_pseudo_fill(_p(grad), _p(hidx), I(num), I(c), I(h), I(w), I(self.npart_), I(self.pad_),
                              I(self.trim_), F(0.0))
        return [grad]


class PseudoEntropyPadOp(_Base):

    def __init__(self, pad, npart, addr, device=0, timeit=False):
        super().__init__(device, timeit)
        self.pad_, self.npart_, self.ctx_ = pad, npart, _ctx(addr)

    def forward(self, x):
        num, c, h, w = x.shape
        p = self.pad_
        hidx, h2, dst, src, pcol, pt = self.ctx_.produce_param(c, h, w, p)
        out = self._top(0, (num, c, h + 2 * p, w + 2 * p))
        lib().orc_entropy_pad(_p(x), _p(out), _p(hidx), _p(h2), _p(dst), _p(src), _p(pcol), _p(pt), I(num), I(c),
                              I(h), I(w), I(self.npart_), I(p))
        return [out]

    def backward(self, grad):
        num, c, hp, wp = grad.shape
        p = self.pad_
        h, w = hp - 2 * p, wp - 2 * p
        hidx, h2, dst, src, pcol, pt = self.ctx_.produce_param(c, h, w, p)
        gin = self._top(1, (num, c, h, w))
        lib().orc_entropy_pad_backward(_p(grad.contiguous()), _p(gin), _p(hidx), _p(h2), _p(dst), _p(src), _p(pcol),
                                       _p(pt), I(num), I(c), I(h), I(w), I(self.npart_), I(p))
        return [gin]


class PseudoQuantOp(_Base):

    def __init__(self, channel, bin_num, npart, weight_decay, check_iters, ntop, top_alpha, addr, device=0,
                 timeit=False):
        super().__init__(device, timeit)
        self.channel_, self.bin_num_, self.npart_, self.ntop_ = channel, bin_num, npart, ntop
        self.top_alpha_ = top_alpha
        self.weight_decay_, self.mod_, self.iter_ = weight_decay, check_iters, 0
        self.ctx_ = _ctx(addr)

    def update_weight(self, weight, ncount):
        # pseudo_quant_cuda.cu:97-143 (pseudo_quant_check_weight + pseudo_quant_scale)
        if self.iter_ % self.mod_ != 0 or self.iter_ == 0:
            return
        levels = self.bin_num_
        w, cnt = weight.data, ncount.data
        for i in range(self.channel_):
            j = levels - 1
            while j > 1:
                if cnt[i, j] >= 1e-3:
                    break
                j -= 1
            tmp = w[i, j] - np.float32(np.log(np.float32(levels - j)))
            w[i, j:] = tmp
            if cnt[i, 0] < 1e-3:
                w[i, 0] = w[i, 0] + torch.exp(w[i, 1])
                tmp = torch.log((torch.exp(w[i, 1]) + torch.exp(w[i, 2])) / 2)
                w[i, 1] = tmp
                w[i, 2] = tmp
        cnt.mul_(self.weight_decay_)

    def forward(self, x, weight, count, train):
        if train:
            self.update_weight(weight, count)
            self.iter_ += 1
        num, c, h, w = x.shape
        hidx = self.ctx_.hindex(h, w)
        tab = np.zeros(c * self.bin_num_, np.float32)
        quant = np.zeros(x.numel(), np.int32)
        val = self._top(0, x.shape)
        idx = self._top(1, x.shape) if self.ntop_ > 1 else None
        # count_data_: per-call histogram, -1 per valid element (pseudo_quant_cuda.cu:12,64,83,167)
        self.count_data_ = torch.zeros((c, self.bin_num_), dtype=torch.float32)
        lib().orc_quant_forward(_p(x), _p(weight.detach().contiguous()), _p(tab), _p(quant), _p(val), _p(idx),
                                _p(self.count_data_), _p(hidx), I(num), I(c), I(h), I(w), I(self.bin_num_),
                                I(self.npart_))
        self.tab_, self.quant_, self.top_alpha_used_ = tab, quant, None
        return [val, idx] if idx is not None else [val]

    def backward(self, grads, x, out):
        # pseudo_quant_cuda.cu:197-311
        num, c, h, w = x.shape
        hidx = self.ctx_.hindex(h, w)
        g_val = grads[0].contiguous()
        g_idx = grads[1].contiguous() if self.ntop_ > 1 and len(grads) > 1 and grads[1] is not None else None
        g_in = self._top("g_in", x.shape)
        g_w = self._top("g_w", (c, self.bin_num_))
        lib().orc_quant_backward(_p(x.contiguous()), _p(out.contiguous()), _p(self.quant_), _p(g_val), _p(g_idx),
                                 _p(self.tab_), _p(g_in), _p(g_w), _p(hidx), F(self.top_alpha_), I(num), I(c), I(h),
                                 I(w), I(self.bin_num_), I(self.npart_))
        return [g_in, g_w, self.count_data_]


class PseudoDQuantOp(_Base):

    def __init__(self, npart, channel, bin_num, addr, device=0, timeit=False):
        super().__init__(device, timeit)
        self.npart_, self.nchannel_, self.bin_num_ = npart, channel, bin_num
        self.ctx_ = _ctx(addr)

    def forward(self, x, weight):
        num, c, h, w = x.shape
        hidx = self.ctx_.hindex(h, w)
        tab = np.zeros(self.nchannel_ * self.bin_num_, np.float32)
        out = self._top(0, x.shape)
        lib().orc_dquant_forward(_p(x), _p(weight.detach().contiguous()), _p(tab), _p(out), _p(hidx), I(num), I(c),
                                 I(h), I(w), I(self.nchannel_), I(self.bin_num_), I(self.npart_))
        return [out]


class ProjectsOp(_Base):

    def __init__(self, h_out, w_out, thetas, phis, fov=0.33333, near=False, device=0, timeit=False):
        super().__init__(device, timeit)
        self.h_out_, self.w_out_ = h_out, w_out
        self.theta_, self.phi_ = np.asarray(list(thetas), np.float32), np.asarray(list(phis), np.float32)
        self.nv_ = len(self.theta_)
        self.fov_, self.near_ = fov, near
        self.tf = {}

    def forward(self, x):
        n, c, h, w = x.shape
        if (h, w) not in self.tf:
            tf = np.zeros(self.nv_ * self.h_out_ * self.w_out_ * 2, np.float32)
            lib().orc_projects_table(_p(self.theta_), _p(self.phi_), I(self.nv_), F(self.fov_), I(self.h_out_),
                                     I(self.w_out_), I(h), I(w), _p(tf))
            self.tf[(h, w)] = tf
        out = self._top(0, (n * self.nv_, c, self.h_out_, self.w_out_))
        lib().orc_projects_forward(_p(x), _p(self.tf[(h, w)]), _p(out), I(n), I(c), I(h), I(w), I(self.nv_),
                                   I(self.h_out_), I(self.w_out_), I(1 if self.near_ else 0))
        self.shape_ = (n, c, h, w)
        return [out]

    def backward(self, grad):
        n, c, h, w = self.shape_
        gin, cnt = self._top(1, (n, c, h, w)), self._top(2, (n, c, h, w))
        lib().orc_projects_backward(_p(gin), _p(cnt), _p(self.tf[(h, w)]), _p(grad.contiguous()), I(n), I(c), I(h), I(w),
                                    I(self.nv_), I(self.h_out_), I(self.w_out_), I(1 if self.near_ else 0))
        return [gin, cnt]


# -- entropy wavefront ops ------------------------------------------------------------
class _Stepped(_Base):

    def __init__(self, device=0, timeit=False):
        super().__init__(device, timeit)
        self.pidx_ = 0

    def restart(self):
        self.pidx_ = 0


def _window(psum, nchannel, h_out, w_out, start_idx):
    st = 0 if psum - nchannel + 1 < 0 else psum - nchannel + 1
    end = psum + 1 if psum < h_out + w_out - 2 else h_out + w_out - 1
    return st, int(start_idx[end] - start_idx[st]) if st < len(start_idx) else 0


class DInput2Op(_Stepped):

    def __init__(self, nchannel, npart, pad, bias, replicate, ctx_addr, device=0, timeit=False):
        super().__init__(device, timeit)
        self.channel_, self.npart_, self.pad_, self.bias_, self.rep_ = nchannel, npart, pad, bias, replicate
        self.ctx_ = _ctx(ctx_addr)

    def forward(self, x):
        num, height, width = x.shape[0] * self.npart_, x.shape[2] // self.npart_, x.shape[3]
        if self._reshape(num, self.channel_, height, width):
            self.pidx_ = 0
        h_out, w_out = height * self.npart_, width
        mod = h_out + w_out + self.channel_ - 2
        index, start_idx = self.ctx_.produce_param_group(height, width)
        p = self.pad_
        top = self._top(0, (self.rep_ * num, self.channel_, height + 2 * p, width + 2 * p))
        psum = self.pidx_
        self.pidx_ += 1
        stride_out = num * self.channel_ * (width + 2 * p) * (height + 2 * p)
        if psum == 0:
            top.zero_()
        elif psum <= mod:
            psum -= 1
            st, len_idx = _window(psum, self.channel_, h_out, w_out, start_idx)
            count = len_idx * num // self.npart_
            if count > 0:
                lib().orc_dinput2(_p(x), _p(index), _p(top), I(count), I(int(start_idx[st])), I(len_idx), I(height),
                                  I(width), I(self.channel_), I(self.npart_), I(psum), I(p), F(self.bias_),
                                  I(self.rep_), L(stride_out))
        return [top]


class EntropyCtxPadRun2Op(_Stepped):

    def __init__(self, pad, npart, ngroup, input, ctx_addr, device=0, timeit=False):
        super().__init__(device, timeit)
        self.pad_, self.npart_, self.ngroup_, self.input_ = pad, npart, ngroup, input
        self.ctx_ = _ctx(ctx_addr)

    def forward(self, x):
        p = self.pad_
        num, channel, height, width = x.shape[0], x.shape[1], x.shape[2] - 2 * p, x.shape[3] - 2 * p
        if self._reshape(num, channel, height, width):
            self.pidx_ = 0
        h_out, w_out = height + 2 * p, width + 2 * p
        cpn = channel // self.ngroup_
        mod = height * self.npart_ + width + p + self.ngroup_ - 2
        n_out = num // self.npart_
        hidx, h2, dst, src, pcol, pt, lst, id_list = self.ctx_.produce_param(channel, height, width, p)
        psum = self.pidx_
        self.pidx_ += 1
        if self.input_:
            psum -= 1
        if 0 <= psum < mod:
            st = 0 if psum - self.ngroup_ + 1 < 0 else psum - self.ngroup_ + 1
            end = psum + 1 if psum < height * self.npart_ + width + p - 2 else height * self.npart_ + width + p - 1
            ntile = int(id_list[end] - id_list[st])
            if ntile > 0:
                count = n_out * cpn * ntile
                lib().orc_ctx_pad_run2(_p(x), _p(dst), _p(src), _p(pcol), _p(pt), _p(lst), _p(hidx), _p(h2), I(count),
                                       I(psum), I(int(id_list[st])), I(ntile), I(cpn), L(h_out * w_out),
                                       L(h_out * w_out * channel * self.npart_), I(width), I(p))
        return [x]


class EntropyConv2Op(_Stepped):

    def __init__(self, npart, channel, ngroup, nout, kernel_size, constrain, pad_in, pad_out, ctx_addr, device=0,
                 timeit=False):
        super().__init__(device, timeit)
        self.npart_, self.channel_, self.ngroup_, self.nout_ = npart, channel, ngroup, nout
        self.kernel_size_, self.constrain_, self.pad_in_, self.pad_out_ = kernel_size, constrain, pad_in, pad_out
        self.group_in_, self.group_out_ = channel // ngroup, nout // ngroup
        self.ctx_ = _ctx(ctx_addr)

    def _run(self, x, weight, bias, act, nset):
        pi, po = self.pad_in_, self.pad_out_
        num, channel, height, width = x.shape[0], x.shape[1], x.shape[2] - 2 * pi, x.shape[3] - 2 * pi
        if self._reshape(num, channel, height, width):
            self.pidx_ = 0
        num_out = num // self.npart_
        h_out, w_out = height * self.npart_, width
        mod = h_out + w_out + self.ngroup_ - 2
        index, start_idx = self.ctx_.produce_param_group(height, width)
        top = self._top(0, (num, self.nout_, height + 2 * po, width + 2 * po))
        psum = self.pidx_
        self.pidx_ += 1
        if psum < mod:
            st, len_idx = _window(psum, self.ngroup_, h_out, w_out, start_idx)
            if len_idx > 0:
                if psum == 0:
                    top.zero_()
                lib().orc_entropy_conv(_p(x), _p(weight.detach().contiguous()), _p(bias.detach().contiguous()),
                                       _p(act.detach().contiguous()) if act is not None else None, _p(top),
                                       _p(index), I(self.kernel_size_), I(self.group_in_), I(self.group_out_),
                                       I(height), I(width), I(int(start_idx[st])), I(psum), I(len_idx), I(channel),
                                       I(self.nout_), I(self.npart_), I(pi), I(po), I(self.constrain_), I(num_out),
                                       I(max(num_out // nset, 1)), I(CONV_ORDER))
        return [top]

    def forward(self, x, w, b):
        return self._run(x, w, b, None, 1)

    def forward_act(self, x, w, b, a):
        return self._run(x, w, b, a, 1)

    def forward_batch(self, x, w, b):
        return self._run(x, w, b, None, w.shape[0])

    def forward_act_batch(self, x, w, b, a):
        return self._run(x, w, b, a, w.shape[0])


class EntropyAddOp(_Stepped):

    def __init__(self, npart, channel, ngroup, pad, ctx_addr, device=0, timeit=False):
        super().__init__(device, timeit)
        self.npart_, self.channel_, self.ngroup_, self.pad_ = npart, channel, ngroup, pad
        self.cpg_ = channel // ngroup
        self.ctx_ = _ctx(ctx_addr)

    def forward(self, x, y):
        p = self.pad_
        num, channel, height, width = x.shape[0], x.shape[1], x.shape[2] - 2 * p, x.shape[3] - 2 * p
        if self._reshape(num, channel, height, width):
            self.pidx_ = 0
        num_out = num // self.npart_
        h_out, w_out = height * self.npart_, width
        mod = h_out + w_out + self.ngroup_ - 2
        index, start_idx = self.ctx_.produce_param_group(height, width)
        psum = self.pidx_
        self.pidx_ += 1
        if psum <= mod:
            st, len_idx = _window(psum, self.ngroup_, h_out, w_out, start_idx)
            count = self.cpg_ * len_idx * num_out
            if count > 0:
                lib().orc_entropy_add(_p(x), _p(y), _p(index), I(count), I(self.cpg_), I(int(start_idx[st])), I(psum),
                                      I(height), I(width), I(self.channel_), I(num_out), I(self.npart_), I(p),
                                      I(len_idx))
        return [x]


class DExtract2Op(_Stepped):

    def __init__(self, npart, nchannel, label, ctx_addr, device=0, timeit=False):
        super().__init__(device, timeit)
        self.npart_, self.nchannel_, self.label_ = npart, nchannel, label
        self.ctx_ = _ctx(ctx_addr)
        self.top_num_ = torch.zeros(1, dtype=torch.int32)

    def _prep(self, x):
        num, channel, height, width = x.shape
        if self._reshape(num, channel, height, width):
            self.pidx_ = 0
            self.top_num_ = torch.zeros(1, dtype=torch.int32)
        self.h_out_, self.w_out_ = height * self.npart_, width
        self.cpn_ = channel // self.nchannel_
        self.mod_ = self.h_out_ + self.w_out_ + self.nchannel_ - 2
        index, start_idx = self.ctx_.produce_param_group(height, width)
        top = self._top(0, (num // self.npart_, self.cpn_, self.h_out_, self.w_out_))
        return num, channel, height, width, index, start_idx, top

    def forward(self, x):
        num, channel, height, width, index, start_idx, top = self._prep(x)
        psum = self.pidx_
        self.pidx_ += 1
        run = False
        if self.label_:
            run = psum < self.mod_
        elif psum == 0:
            top.zero_()
        elif psum <= self.mod_:
            psum -= 1
            run = True
        if run:
            st, len_idx = _window(psum, self.nchannel_, self.h_out_, self.w_out_, start_idx)
            count = len_idx * num // self.npart_ * self.cpn_
            self.top_num_[0] = count // self.cpn_
            if count > 0:
                lib().orc_dextract2(_p(x), _p(index), _p(top), I(count), I(int(start_idx[st])), I(len_idx), I(height),
                                    I(width), I(channel), I(self.cpn_), I(self.npart_), I(psum), L(0), I(0))
        return [top, self.top_num_]

    def forward_batch(self, x):
        num, channel, height, width, index, start_idx, top = self._prep(x)
        psum = self.pidx_
        self.pidx_ += 1
        nout = num // self.npart_ // 3
        if psum < self.mod_:
            st, len_idx = _window(psum, self.nchannel_, self.h_out_, self.w_out_, start_idx)
            count = len_idx * num // self.npart_ * self.cpn_
            self.top_num_[0] = nout * len_idx
            if count > 0:
                lib().orc_dextract2(_p(x), _p(index), _p(top), I(count), I(int(start_idx[st])), I(len_idx), I(height),
                                    I(width), I(channel), I(self.cpn_), I(self.npart_), I(psum),
                                    L(self.cpn_ * self.h_out_ * self.w_out_ * nout), I(len_idx * self.cpn_ * nout))
        return [top, self.top_num_]


class EntropyGmmTableOp(_Base):

    def __init__(self, nstep, bias, num_gaussian, total_region, beta=1e-6, device=0, timeit=False):
        super().__init__(device, timeit)
        self.nstep_, self.bias_, self.num_gaussian_ = nstep, bias, num_gaussian
        self.total_region_, self.beta_ = total_region, beta

    def forward(self, weight, delta, mean, tnum):
        rows = weight.numel() // self.num_gaussian_
        if weight.dim() == 4:
            rows = weight.shape[0] * weight.shape[2] * weight.shape[3]
        top = self._top(0, (rows, self.nstep_ + 1))
        tn = int(tnum[0])
        lib().orc_gmm_table(_p(weight), _p(delta), _p(mean), _p(top), I(tn), I(self.num_gaussian_), I(self.nstep_),
                            F(self.bias_), F(self.total_region_), F(self.beta_), I(0))
        return [top]

    def forward_batch(self, data, tnum):
        stride = data.numel() // 3
        top = self._top(0, (data.shape[0] * data.shape[2] * data.shape[3] // 3, self.nstep_ + 1))
        tn = int(tnum[0])
        if tn > 0:
            flat = data.view(-1)
            lib().orc_gmm_table(_p(flat[0:]), _p(flat[stride:]), _p(flat[2 * stride:]), _p(top), I(tn),
                                I(self.num_gaussian_), I(self.nstep_), F(self.bias_), F(self.total_region_),
                                F(self.beta_), I(1))
        return [top]


def tile_conv2d(owner, x, weight, bias, stride, slope=None, col_limit=None, npart=0):
    """1e-4 reference of the dense tile convolution: torch's CPU conv (oneDNN)."""
    y = torch.nn.functional.conv2d(x, weight, bias, stride)
    return torch.nn.functional.prelu(y, slope) if slope is not None else y


def conv2d_chain(x, weight, bias, stride, slope=None):
    """bit-exact reference: one k-ascending fp32 fmaf chain per output."""
    tn, cin, h, w = x.shape
    cout, _, k, _ = weight.shape
    out = torch.empty((tn, cout, (h - k) // stride + 1, (w - k) // stride + 1), dtype=torch.float32)
    lib().orc_conv2d_chain(_p(x.contiguous()), _p(weight.detach().contiguous()),
                           _p(bias.detach().contiguous()) if bias is not None else None,
                           _p(slope.detach().contiguous()) if slope is not None else None, _p(out), I(tn), I(cin),
                           I(h), I(w), I(cout), I(k), I(stride))
    return out
