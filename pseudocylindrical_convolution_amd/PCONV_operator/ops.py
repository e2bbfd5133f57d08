"""The nn.Module operator surface of the reference (PCONV_operator/*.py), written
against the native op classes of `backend.ops()`.

Class names, constructor signatures, parameter names/shapes (they are part of the
checkpoint format, pseudo_codec.py:223-227) and forward semantics follow the
reference files cited on each class.
"""
import math

import torch
from torch import nn

from . import backend
from .BaseOpModule import BaseOpModule, native_call
from .GDN import LowerBound
from .base import set_weight


def _per_gpu(module, factory):
    """{gpu id: native op} for every id in module.device_list"""
    return {gid: factory(gid) for gid in module.device_list}


# --------------------------------------------------------------------------
# sphere <-> tile stack, pixel shuffle, viewports
# --------------------------------------------------------------------------
class SphereSlice(BaseOpModule):
    """ERP image -> stack of latitude tiles (reference: SphereSlice.py:27-37)."""

    def __init__(self, npart, interp_type=0, pad=0, opt=False, device=0, time_it=False):
        super(SphereSlice, self).__init__(device)
        weight = set_weight(npart, opt)
        self.op = _per_gpu(self, lambda g: backend.ops().SphereSliceOp(npart, interp_type, pad, weight, g, time_it))

    def forward(self, x):
        return native_call(self, 'forward', (x,))


class SphereUslice(BaseOpModule):
    """tile stack -> ERP image (reference: SphereUslice.py:25-34)."""

    def __init__(self, npart, interp_type=0, pad=0, opt=False, device=0, time_it=False):
        super(SphereUslice, self).__init__(device)
        weight = set_weight(npart, opt)
        self.op = _per_gpu(self, lambda g: backend.ops().SphereUsliceOp(npart, interp_type, pad, weight, g, time_it))

    def forward(self, x):
        return native_call(self, 'forward', (x,))


class Dtow(BaseOpModule):
    """depth <-> width pixel shuffle (reference: Dtow.py:24-34)."""

    def __init__(self, stride=2, d2w=False, device=0, time_it=False):
        super(Dtow, self).__init__(device)
        self.op = _per_gpu(self, lambda g: backend.ops().DtowOp(stride, d2w, g, time_it))

    def forward(self, x):
        return native_call(self, 'forward', (x,))


class MultiProjectM(BaseOpModule):
    """rectilinear viewports at given (theta, phi) (reference: MultiProject.py:24-33)."""

    def __init__(self, h, w, thetas, phis, fov=0.6, near=False, device_id=0, time_flag=False):
        super(MultiProjectM, self).__init__(device_id)
        self.op = _per_gpu(self, lambda g: backend.ops().ProjectsOp(int(h), int(w), thetas, phis, fov, near, g, time_flag))

    def forward(self, x):
        return native_call(self, 'forward', (x,))


class MultiProject(MultiProjectM):
    """the 14 evaluation viewports of the paper (reference: MultiProject.py:35-45)."""
    THETAS = [-0.5, 0, 0.5, 1, -0.5, 0, 0.5, 1, -0.5, 0, 0.5, 1, 0, 0]
    PHIS = [0, 0, 0, 0, 0.25, 0.25, 0.25, 0.25, -0.25, -0.25, -0.25, -0.25, 0.5, -0.5]

    def __init__(self, h, w, fov=0.6, near=False, device_id=0, time_flag=False):
        self.thetas, self.phis = list(self.THETAS), list(self.PHIS)
        super(MultiProject, self).__init__(h, w, self.thetas, self.phis, fov, near, device_id, time_flag)


# --------------------------------------------------------------------------
# pseudocylindrical padding / masking / quantisation
# --------------------------------------------------------------------------
class _Context(BaseOpModule):

    def setup_context(self, w):
        for op in self.op.values():
            op.start_context(w)

    def get_addr(self, gid):
        return self.op[gid].addr()


class PseudoContextV2(_Context):
    """shared tile-geometry cache (reference: PseudoContextV2.py:13-28)."""

    def __init__(self, npart, opt=True, rt=20, device=0, time_it=False):
        super(PseudoContextV2, self).__init__(device)
        weight = set_weight(npart, opt)
        self.op = _per_gpu(self, lambda g: backend.ops().PseudoContextOp(npart, rt, weight, g, time_it))

    def produce_fill_param(self, gid, h, w):
        return self.op[gid].produce_fill_param(h, w)


class PseudoEntropyContext(_Context):
    """(reference: PseudoContextV2.py:30-42) -- training-time entropy net geometry."""

    def __init__(self, npart, context_version=1, opt=True, rt=20, device=0, time_it=False):
        super(PseudoEntropyContext, self).__init__(device)
        weight = set_weight(npart, opt)
        self.op = _per_gpu(self, lambda g: backend.ops().PseudoEntropyContextOp(npart, rt, context_version, weight, g, time_it))


class EntropyContextNew(_Context):
    """wavefront schedule + causal halo lists (reference: EntropyContextNew.py:8-21)."""

    def __init__(self, npart, rt=18, opt=False, device=0, time_it=False):
        super(EntropyContextNew, self).__init__(device)
        weight = set_weight(npart, opt)
        self.op = _per_gpu(self, lambda g: backend.ops().EntropyContextOp(npart, rt, weight, g, time_it))


class PseudoPadV2(BaseOpModule):
    """halo from the neighbouring tiles + circular wrap (reference: PseudoContextV2.py:89-97)."""

    def __init__(self, pad, npart, ctx, device=0, time_it=False):
        super(PseudoPadV2, self).__init__(device)
        self.op = _per_gpu(self, lambda g: backend.ops().PseudoPadOp(pad, npart, ctx.get_addr(g), g, time_it))

    def forward(self, x):
        op = self.native(x)
        if getattr(x, "_pconv_ring", None) is not None and hasattr(op, "forward_ring") \
                and not torch.is_grad_enabled():
            # x already lives inside a padded buffer (its producer wrote it there):
            # only the ring is computed, no copy
            return op.forward_ring(x)
        return native_call(self, 'forward', (x,))


class PseudoEntropyPad(BaseOpModule):
    """(reference: PseudoContextV2.py:61-69) -- training-time causal pad."""

    def __init__(self, pad, npart, ctx, device=0, time_it=False):
        super(PseudoEntropyPad, self).__init__(device)
        self.op = _per_gpu(self, lambda g: backend.ops().PseudoEntropyPadOp(pad, npart, ctx.get_addr(g), g, time_it))

    def forward(self, x):
        return native_call(self, 'forward', (x,))


class PseudoFillV2(BaseOpModule):
    """in-place constant fill outside each tile's valid region
    (reference: PseudoContextV2.py:116-129); the context flavour picks version 0/1/2."""

    def __init__(self, pad, npart, ctx, fvalue=0, trim=0, device=0, time_it=False):
        super(PseudoFillV2, self).__init__(device)
        version = 0 if isinstance(ctx, PseudoContextV2) else (1 if isinstance(ctx, PseudoEntropyContext) else 2)
        self.op = _per_gpu(self, lambda g: backend.ops().PseudoFillOp(pad, npart, fvalue, trim, ctx.get_addr(g), version, g, time_it))

    def forward(self, x):
        return native_call(self, 'forward', (x,))


class PseudoGDNV2(nn.Module):
    """masked generalised divisive normalisation (reference: PseudoContextV2.py:133-216):
    y = x / sqrt(beta + gamma * x^2) inside the valid columns, identity elsewhere."""

    def __init__(self, ch, npart, ctx, device=0, inverse=False, beta_min=1e-6, gamma_init=.1,
                 reparam_offset=2 ** -18):
        super(PseudoGDNV2, self).__init__()
        self.inverse = inverse
        self.beta_min = beta_min
        self.gamma_init = gamma_init
        self.reparam_offset = torch.FloatTensor([reparam_offset])
        self.trim = PseudoFillV2(0, npart, ctx, device=device)
        self.mask = None
        self.__dict__["ctx"] = ctx  # not a sub-module: owned by the codec
        gid = device if isinstance(device, int) else device[0]
        self.build(ch, torch.device(backend.device_of(gid)))

    def build(self, ch, device):
        self.pedestal = self.reparam_offset ** 2
        self.beta_bound = (self.beta_min + self.reparam_offset ** 2) ** .5
        self.gamma_bound = self.reparam_offset
        self.beta = nn.Parameter(torch.sqrt(torch.ones(ch) + self.pedestal).to(device))
        gamma = torch.sqrt(self.gamma_init * torch.eye(ch) + self.pedestal)
        self.gamma = nn.Parameter(gamma.to(device))
        self.pedestal = self.pedestal.to(device)

    def setup_mask(self, x):
        if self.mask is not None and self.mask.shape == x.shape and self.mask.device == x.device:
            return
        self.mask = self.trim(torch.ones_like(x).detach())

    def effective(self):
        """(gamma, beta) after the lower-bound re-parametrisation, cached until a
        parameter changes"""
        key = (self.gamma._version, self.beta._version, self.gamma.device, backend.param_epoch())
        if getattr(self, "_effective", (None,))[0] != key:
            with torch.no_grad():
                pedestal = self.pedestal.to(self.gamma.device)
                beta = LowerBound.apply(self.beta, self.beta_bound) ** 2 - pedestal
                gamma = LowerBound.apply(self.gamma, self.gamma_bound) ** 2 - pedestal
            self._effective = (key, gamma.contiguous(), beta.contiguous())
        return self._effective[1], self._effective[2]

    def forward(self, inputs, residual=None, trim=None, ring=0):
        """reference signature: forward(inputs).  residual / trim (a PseudoFill module):
        the `trim(residual + gdn(inputs))` that ends ResidualBlockDown / ResidualBlockUp,
        evaluated in the same launch when the backend can."""
        ops = backend.ops()
        if hasattr(ops, "tile_gdn") and inputs.is_cuda and not torch.is_grad_enabled():
            # one launch on the tile-convolution kernel (squares, 1x1 MFMA GEMM, sqrt,
            # divide, residual and the valid-column mask fused)
            gamma, beta = self.effective()
            ctx_op = self.ctx.native(inputs)
            limit, npart = ops.conv_col_limit(ctx_op, inputs.shape[2], inputs.shape[3], 0, inputs)
            return ops.tile_gdn(self, inputs, gamma, beta, self.inverse, limit, npart, residual, ring)
        out = self._formula(inputs)
        if residual is not None:
            out = residual + out
        return trim(out) if trim is not None else out

    def _formula(self, inputs):
        self.pedestal = self.pedestal.to(inputs.device)
        ch = inputs.size(1)
        self.setup_mask(inputs)
        inputs = inputs * self.mask
        beta = LowerBound.apply(self.beta, self.beta_bound) ** 2 - self.pedestal
        gamma = LowerBound.apply(self.gamma, self.gamma_bound) ** 2 - self.pedestal
        norm_ = nn.functional.conv2d(inputs ** 2, gamma.view(ch, ch, 1, 1), beta)
        norm_ = torch.sqrt(norm_)
        norm_ = norm_ * self.mask + 1 - self.mask
        return inputs * norm_ if self.inverse else inputs / norm_


class _QuantCall(torch.autograd.Function):
    """autograd bridge of the quantiser (reference: PseudoContextV2.py:218-239): backward hands
    back the input gradient, the level-table gradient and the per-call histogram for `count`"""

    @staticmethod
    def forward(ctx, module, x, weight, count, training):
        op = module.native(x)
        outs = op.forward(x, weight, count, training)
        ctx.op = op
        ctx.save_for_backward(x, outs[0])
        return outs[0] if len(outs) == 1 else (outs[0], outs[1])

    @staticmethod
    def backward(ctx, *grads):
        x, out = ctx.saved_tensors
        g = [t.contiguous() if t is not None else torch.zeros_like(x) for t in grads]
        gx, gw, gc = ctx.op.backward(g, x, out)
        return None, gx, gw, gc.clone().detach(), None


class PseudoQUANTV2(BaseOpModule):
    """learned 8-level per-channel quantiser (reference: PseudoContextV2.py:241-255)."""

    def __init__(self, channel, bin_num, npart, ctx, check_iters=100, weight_decay=0.9, ntop=1,
                 top_alpha=0.1, device_id=0, time_flag=False):
        super(PseudoQUANTV2, self).__init__(device_id)
        dev = backend.device_of(self.device_list[0])
        first = 1. / (bin_num + 1)
        weight = torch.zeros((channel, bin_num), dtype=torch.float32)
        weight[:, 0] = first
        weight[:, 1:] = math.log(first)
        self.weight = nn.Parameter(weight.to(dev))
        self.count = nn.Parameter(torch.zeros((channel, bin_num), dtype=torch.float32).to(dev))
        self.ntop = ntop
        self.op = _per_gpu(self, lambda g: backend.ops().PseudoQuantOp(
            channel, bin_num, npart, weight_decay, check_iters, ntop, top_alpha, ctx.get_addr(g), g, time_flag))

    def forward(self, x):
        if not x.is_contiguous():
            x = x.contiguous()
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad):
            return _QuantCall.apply(self, x, self.weight, self.count, self.training)
        outs = self.native(x).forward(x, self.weight, self.count, self.training)
        return outs[0] if len(outs) == 1 else (outs[0], outs[1])


class PseudoDQUANT(BaseOpModule):
    """index -> level value (reference: PseudoContextV2.py:271-280)."""

    def __init__(self, channel, bin_num, npart, ctx, device_id=0, time_flag=False):
        super(PseudoDQUANT, self).__init__(device_id)
        dev = backend.device_of(self.device_list[0])
        self.weight = nn.Parameter(torch.zeros((channel, bin_num), dtype=torch.float32).to(dev))
        self.op = _per_gpu(self, lambda g: backend.ops().PseudoDQuantOp(npart, channel, bin_num, ctx.get_addr(g), g, time_flag))

    def forward(self, x):
        if not x.is_contiguous():
            x = x.contiguous()
        return self.native(x).forward(x, self.weight)[0]


# --------------------------------------------------------------------------
# entropy wavefront
# --------------------------------------------------------------------------
class _Restartable(BaseOpModule):

    def restart(self):
        for op in self.op.values():
            op.restart()


class EntropyAdd(_Restartable):
    """in-place residual add at the current wavefront (reference: EntropyContextNew.py:36-48)."""

    def __init__(self, npart, channel, ngroup, pad, ctx, device=0, time_it=False):
        super(EntropyAdd, self).__init__(device)
        self.op = _per_gpu(self, lambda g: backend.ops().EntropyAddOp(npart, channel, ngroup, pad, ctx.get_addr(g), g, time_it))

    def forward(self, x, y):
        return self.native(x).forward(x, y)[0]


class EntropyCtxPadRun2(_Restartable):
    """in-place causal halo update (reference: EntropyContextNew.py:63-75)."""

    def __init__(self, pad, npart, ngroup, ctx, input=False, device=0, time_it=False):
        super(EntropyCtxPadRun2, self).__init__(device)
        self.op = _per_gpu(self, lambda g: backend.ops().EntropyCtxPadRun2Op(pad, npart, ngroup, input, ctx.get_addr(g), g, time_it))

    def forward(self, x):
        return self.native(x).forward(x)[0]


class DExtract2(_Restartable):
    """gather the current wavefront into a packed list (reference: EntropyContextNew.py:91-103)."""

    def __init__(self, npart, nchannel, label, ctx, device=0, time_it=False):
        super(DExtract2, self).__init__(device)
        self.op = _per_gpu(self, lambda g: backend.ops().DExtract2Op(npart, nchannel, label, ctx.get_addr(g), g, time_it))

    def forward(self, x):
        if not x.is_contiguous():
            x = x.contiguous()
        out = self.native(x).forward(x)
        return out[0], out[1]


class DExtract2Batch(_Restartable):
    """packed GMM parameters of the three sub-networks (reference: EntropyContextNew.py:120-132)."""

    def __init__(self, npart, nchannel, ctx, device=0, time_it=False):
        super(DExtract2Batch, self).__init__(device)
        self.op = _per_gpu(self, lambda g: backend.ops().DExtract2Op(npart, nchannel, True, ctx.get_addr(g), g, time_it))

    def forward(self, x):
        out = self.native(x).forward_batch(x)
        return out[0], out[1]


class DInput2(_Restartable):
    """scatter decoded symbols into the padded context tensor (reference: EntropyContextNew.py:149-161)."""

    def __init__(self, nchannel, npart, ctx, pad=0, bias=0, repeat=1, device=0, time_it=False):
        super(DInput2, self).__init__(device)
        self.op = _per_gpu(self, lambda g: backend.ops().DInput2Op(nchannel, npart, pad, bias, repeat, ctx.get_addr(g), g, time_it))

    def forward(self, x):
        if not x.is_contiguous():
            x = x.contiguous()
        return self.native(x).forward(x)[0]


class EntropyConv2(_Restartable):
    """masked grouped conv at wavefront positions (reference: EntropyContextNew.py:214-236)."""

    def __init__(self, npart, ngroup, c_in, c_out, kernel_size, ctx, pad_in=2, pad_out=2, hidden=False,
                 act=True, device=0, time_it=False):
        super(EntropyConv2, self).__init__(device)
        constrain = 6 if hidden else 5
        channel, nout = ngroup * c_in, ngroup * c_out
        self.op = _per_gpu(self, lambda g: backend.ops().EntropyConv2Op(
            npart, channel, ngroup, nout, kernel_size, constrain, pad_in, pad_out, ctx.get_addr(g), g, time_it))
        self.weight = nn.Parameter(torch.rand((nout, channel, kernel_size, kernel_size), dtype=torch.float32))
        self.bias = nn.Parameter(torch.zeros((nout), dtype=torch.float32))
        self.act = act
        self.relu = nn.Parameter(torch.zeros((nout), dtype=torch.float32)) if act else None

    def forward(self, x):
        op = self.native(x)
        if self.act:
            return op.forward_act(x, self.weight, self.bias, self.relu)[0]
        return op.forward(x, self.weight, self.bias)[0]


class EntropyConv2Batch(_Restartable):
    """`batch` independent weight sets over a replica-major batch
    (reference: EntropyContextNew.py:238-259)."""

    def __init__(self, npart, ngroup, c_in, c_out, kernel_size, ctx, pad_in=2, pad_out=2, batch=3,
                 hidden=False, act=True, device=0, time_it=False):
        super(EntropyConv2Batch, self).__init__(device)
        constrain = 6 if hidden else 5
        channel, nout = ngroup * c_in, ngroup * c_out
        self.op = _per_gpu(self, lambda g: backend.ops().EntropyConv2Op(
            npart, channel, ngroup, nout, kernel_size, constrain, pad_in, pad_out, ctx.get_addr(g), g, time_it))
        self.weight = nn.Parameter(torch.rand((batch, nout, channel, kernel_size, kernel_size), dtype=torch.float32))
        self.bias = nn.Parameter(torch.rand((batch, nout), dtype=torch.float32))
        self.act = act
        self.relu = nn.Parameter(torch.rand((batch, nout), dtype=torch.float32)) if act else None

    def forward(self, x):
        op = self.native(x)
        if self.act:
            return op.forward_act_batch(x, self.weight, self.bias, self.relu)[0]
        return op.forward_batch(x, self.weight, self.bias)[0]


class EntropyConvD(nn.Module):
    """causal pad + masked conv (reference: EntropyContextNew.py:262-273)."""

    def __init__(self, ngroups, cin, cout, hidden, npart, out_layer, ctx, device_id, act=True):
        super(EntropyConvD, self).__init__()
        self.pad = EntropyCtxPadRun2(2, npart, ngroups, ctx, not hidden, device=device_id)
        self.conv = EntropyConv2(npart, ngroups, cin, cout, 5, ctx, 2, 0 if out_layer else 2, hidden=hidden,
                                 act=act, device=device_id)

    def forward(self, x):
        return self.conv(self.pad(x))


class EntropyResidualBlockD(nn.Module):
    """(reference: EntropyContextNew.py:275-286)"""

    def __init__(self, ngroups, cpn, npart, ctx, device_id=0):
        super(EntropyResidualBlockD, self).__init__()
        self.conv1 = EntropyConvD(ngroups, cpn, cpn, True, npart, False, ctx, device_id, True)
        self.conv2 = EntropyConvD(ngroups, cpn, cpn, True, npart, False, ctx, device_id, True)
        self.add = EntropyAdd(npart, cpn * ngroups, ngroups, 2, ctx, device=device_id)

    def forward(self, x):
        return self.add(self.conv2(self.conv1(x)), x)


class EntropyGmmTable(BaseOpModule):
    """GMM parameters -> integer CDF rows (reference: EntropyGmmTable.py:24-33)."""

    def __init__(self, nstep, bias, num_gaussian, total_region, beta=1e-6, device=0, time_it=False):
        super(EntropyGmmTable, self).__init__(device)
        self.op = _per_gpu(self, lambda g: backend.ops().EntropyGmmTableOp(nstep, bias, num_gaussian, total_region, beta, g, time_it))

    def forward(self, weight, delta, mean, ntop):
        args = [t if t.is_contiguous() else t.contiguous() for t in (weight, delta, mean)]
        return self.native(weight).forward(args[0], args[1], args[2], ntop)[0]


class EntropyBatchGmmTable(BaseOpModule):
    """the packed (weights | deltas | means) variant used by the codec
    (reference: EntropyGmmTable.py:49-57)."""

    def __init__(self, nstep, bias, num_gaussian, total_region, beta=1e-6, device=0, time_it=False):
        super(EntropyBatchGmmTable, self).__init__(device)
        self.op = _per_gpu(self, lambda g: backend.ops().EntropyGmmTableOp(nstep, bias, num_gaussian, total_region, beta, g, time_it))

    def forward(self, x, ntop):
        if not x.is_contiguous():
            x = x.contiguous()
        return self.native(x).forward_batch(x, ntop)[0]


class EntropyGmm(BaseOpModule):
    """rate of a label under a GMM, -log sum_i w_i (Phi(b) - Phi(a))
    (reference: EntropyGmm.py:27-36)."""

    def __init__(self, num_gaussian=3, ignore_label=0, device=0, time_it=False):
        super(EntropyGmm, self).__init__(device)
        self.op = _per_gpu(self, lambda g: backend.ops().EntropyGmmOp(num_gaussian, ignore_label, g, time_it))

    def forward(self, weight, delta, mean, label):
        return native_call(self, 'forward', (weight, delta, mean, label))


class ContextReshape(BaseOpModule):
    """(N, G*cpg, H, W) -> (N*G*H*W, cpg) (reference: ContextReshape.py:24-31)."""

    def __init__(self, ngroup, device=0, time_it=False):
        super(ContextReshape, self).__init__(device)
        self.op = _per_gpu(self, lambda g: backend.ops().ContextReshapeOp(ngroup, g, time_it))

    def forward(self, x):
        return native_call(self, 'forward', (x,))


class _MaskConv(BaseOpModule):

    def __init__(self, constrain, ngroup, c_in, c_out, kernel_size, device, time_it):
        super(_MaskConv, self).__init__(device)
        self.op = _per_gpu(self, lambda g: backend.ops().MaskConstrainOp(constrain, ngroup, g, time_it))
        self.weight = nn.Parameter(torch.empty((c_out * ngroup, c_in * ngroup, kernel_size, kernel_size), dtype=torch.float32))
        torch.nn.init.kaiming_normal_(self.weight)
        self.bias = nn.Parameter(torch.zeros(c_out * ngroup, dtype=torch.float32))

    def forward(self, x):
        self.native(self.weight.data).forward(self.weight.data)  # zero the non-causal taps in place
        return nn.functional.conv2d(x, self.weight, self.bias)


class MaskConv2(_MaskConv):
    """dense conv whose weight is masked to the 3-D causal neighbourhood
    (reference: MaskConstrain.py:24-38)."""

    def __init__(self, ngroup, c_in, c_out, kernel_size, hidden=False, device=0, time_it=False):
        super(MaskConv2, self).__init__(6 if hidden else 5, ngroup, c_in, c_out, kernel_size, device, time_it)


class MaskConv3(_MaskConv):
    """raster-causal variant (reference: MaskConstrain.py:40-53)."""

    def __init__(self, ngroup, c_in, c_out, kernel_size, hidden=False, device=0, time_it=False):
        super(MaskConv3, self).__init__(2 if hidden else 1, ngroup, c_in, c_out, kernel_size, device, time_it)


# --------------------------------------------------------------------------
# pure-torch helpers
# --------------------------------------------------------------------------
class _ExtractFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, dims):
        ctx.full_shape = x.shape
        return x[:, :dims].contiguous()

    @staticmethod
    def backward(ctx, grad_output):
        full = grad_output.new_zeros(ctx.full_shape)
        full[:, :grad_output.shape[1]] = grad_output
        return full, None


class Extract(nn.Module):
    """keep the first `dims` channels (reference: StubMask.py:43-50)."""

    def __init__(self, dims):
        super(Extract, self).__init__()
        self.dims = dims

    def forward(self, x):
        return _ExtractFn.apply(x, self.dims)


class StubMask(nn.Module):
    """ones on the first `dims` channels, zeros elsewhere (reference: StubMask.py:21-41)."""

    def __init__(self, dims=192):
        super(StubMask, self).__init__()
        self.dims = dims
        self.mask = None

    def setup_mask(self, x):
        if self.mask is not None and self.mask.shape == x.shape:
            if self.mask.device != x.device:
                self.mask = self.mask.to(x.device)
            return
        self.mask = torch.ones_like(x)
        self.mask[:, self.dims:] = 0

    def forward(self, x):
        self.setup_mask(x)
        return self.mask


class _DropGradFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, keep):
        ctx.keep = keep
        return x.clone()

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output * ctx.keep, None


class DropGrad(nn.Module):
    """identity forward, optionally blocks the gradient (reference: DropGrad.py:16-22)."""

    def __init__(self, drop=True):
        super(DropGrad, self).__init__()
        self.drop = 0 if drop else 1

    def forward(self, x):
        return _DropGradFn.apply(x, self.drop)
