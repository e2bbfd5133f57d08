"""Analysis / synthesis transforms of the codec (reference: model_zoo_v2.py:8-211).

Module and parameter names follow the reference so that its checkpoints
(`{idx}_encoder.pt`, `{idx}_decoder.pt`) load with strict=True.  The dense
convolutions are `TileConv2d` modules: an nn.Conv2d (same parameters) whose
forward runs the hand-written fp32-MFMA implicit-GEMM kernel through the active
backend, optionally fused with the PReLU that follows it.
"""
import os

import torch
from torch import nn

from .PCONV_operator import Dtow, PseudoContextV2, PseudoFillV2, PseudoGDNV2, PseudoPadV2, backend

__all__ = ["ClipData", "TileConv2d", "ResidualBlock", "AttentionBlock", "ResidualBlockV2", "ResidualBlockDown",
           "SphereConv2", "EncoderV2", "ResidualBlockUp", "SphereConvOld", "DecoderV2"]


class _LeakyClip(torch.autograd.Function):
    # identity on [0, 1], slope 0.01 outside (reference: model_zoo_v2.py:8-26)

    @staticmethod
    def forward(ctx, x):
        below, above = x < 0, x > 1
        y = x.clone().detach()
        y[below] = x[below] * 0.01
        y[above] = 1 + (x[above] - 1) * 0.01
        ctx.save_for_backward(below, above)
        return y

    @staticmethod
    def backward(ctx, grad_output):
        below, above = ctx.saved_tensors
        g = grad_output.clone().detach()
        g[below] = grad_output[below] * 0.01
        g[above] = grad_output[above] * 0.01
        return g


class ClipData(nn.Module):

    def forward(self, x):
        return _LeakyClip.apply(x)


class TileConv2d(nn.Conv2d):
    """nn.Conv2d (no padding, k in {1,3}, stride in {1,2}) evaluated by the
    backend's tile convolution, with what follows it in the graph folded into the
    kernel's epilogue when the backend can:
      prelu / sigmoid   the activation module that follows
      gate              y = gate * y          (attention: trunk * sigmoid(conv))
      residual          y = residual + y      (x + f(x) of the residual blocks)
      trim              the PseudoFill module that ends the block
      d2w               the Dtow(2, True) module that follows (pixel shuffle done by the store)
      ring              write the result into the interior of a buffer padded by `ring`,
                        so that the PseudoPad of the consumer only fills the ring (no copy)
    `live=(ctx, base)` tells the kernel which output columns can ever be read: tile t
    only needs columns < widths_t(base) + (wo - base), base = tile width of the
    scale the output lives in; 64-column blocks beyond that are written as zeros
    without being computed (they are dead: every consumer either trims them or
    never reads them).  Without fusion support (CPU oracle backend,
    PCONV_TILE_CONV=vendor for A/B timing -- never used for parity claims) the same
    operations run one by one, in the same order."""

    def _native(self, x, prelu=None, live=None, sigmoid=False, gate=None, residual=None, trim=None, ring=0,
                d2w=None):
        ops = backend.ops()
        slope = prelu.weight if prelu is not None else None
        vendor = os.environ.get("PCONV_TILE_CONV", "native") == "vendor" or not hasattr(ops, "tile_conv2d")
        fused = (not vendor) and getattr(ops, "FUSED_EPILOGUE", False)
        limit, npart = None, 0
        if live is not None and hasattr(ops, "conv_col_limit") and os.environ.get("PCONV_SKIP_DEAD", "1") == "1":
            ctx, base = live
            k, s = self.kernel_size[0], self.stride[0]
            wo = (x.shape[3] - k) // s + 1
            limit, npart = ops.conv_col_limit(ctx.native(x), x.shape[2], base, wo - base, x)
            if trim is not None and wo != base:
                raise ValueError("trim on an output that still carries halo columns")
        if fused and (trim is None or limit is not None):
            return ops.tile_conv2d(self, x, self.weight, self.bias, self.stride[0], slope, limit, npart,
                                   sigmoid=sigmoid, gate=gate, residual=residual, trim=trim is not None,
                                   ring=ring, d2w=d2w is not None)
        if vendor:
            y = nn.functional.conv2d(x, self.weight, self.bias, self.stride)
            y = nn.functional.prelu(y, slope) if slope is not None else y
        else:
            y = ops.tile_conv2d(self, x, self.weight, self.bias, self.stride[0], slope, limit, npart)
        if sigmoid:
            y = torch.sigmoid(y)
        if gate is not None:
            y = gate * y
        if residual is not None:
            y = residual + y
        if d2w is not None:
            y = d2w(y)
        return trim(y) if trim is not None else y

    def forward(self, x, live=None, **epilogue):
        return self._native(x, None, live, **epilogue)

    def fuse(self, x, prelu, live=None, **epilogue):
        return self._native(x, prelu, live, **epilogue)


# block outputs live inside buffers padded by the largest pad of the graph, so that the
# consumer's PseudoPad fills a ring instead of copying the tensor (PCONV_PAD_RING=0: A/B timing)
RING = 2 if os.environ.get("PCONV_PAD_RING", "1") == "1" else 0


def _conv(cin, cout, k, stride=1):
    return TileConv2d(cin, cout, k, stride)


class ResidualBlock(nn.Module):
    """1x1 -> 3x3 -> 1x1 bottleneck on the padded tile (reference: model_zoo_v2.py:36-53)."""

    def __init__(self, channels, npart, ctx, device_id=0):
        super(ResidualBlock, self).__init__()
        mid = channels // 2
        self.pad = PseudoPadV2(1, npart, ctx, device=device_id)
        self.conv1 = _conv(channels, mid, 1)
        self.relu1 = nn.PReLU(mid)
        self.conv2 = _conv(mid, mid, 3)
        self.relu2 = nn.PReLU(mid)
        self.conv3 = _conv(mid, channels, 1)
        self.trim = PseudoFillV2(0, npart, ctx, device=device_id)
        self.__dict__["ctx"] = ctx  # not a sub-module: the context is owned by the codec

    def forward(self, x):
        live = (self.ctx, x.shape[3])
        y = self.conv1.fuse(self.pad(x), self.relu1, live)
        y = self.conv2.fuse(y, self.relu2, live)
        return self.conv3(y, live, residual=x, trim=self.trim, ring=RING)


class AttentionBlock(nn.Module):
    """x + trunk(x) * sigmoid(attention(x)) (reference: model_zoo_v2.py:55-76)."""

    def __init__(self, channels, npart, ctx, device_id=0):
        super(AttentionBlock, self).__init__()
        block = lambda: ResidualBlock(channels, npart, ctx, device_id)
        self.trunk = nn.Sequential(block(), block(), block())
        self.attention = nn.Sequential(block(), block(), block(), _conv(channels, channels, 1), nn.Sigmoid())
        self.trim = PseudoFillV2(0, npart, ctx, device=device_id)
        self.__dict__["ctx"] = ctx

    def forward(self, x):
        # x + trunk(x) * sigmoid(conv(...)): the sigmoid (attention[4]), the product, the
        # sum and the trim run in the epilogue of the 1x1 conv (attention[3])
        a = x
        for m in self.attention[:3]:
            a = m(a)
        return self.attention[3](a, (self.ctx, x.shape[3]), sigmoid=True, gate=self.trunk(x), residual=x,
                                 trim=self.trim, ring=RING)


class ResidualBlockV2(nn.Module):
    """two 3x3 convs on a 2-pixel halo (reference: model_zoo_v2.py:78-93)."""

    def __init__(self, channels, npart, ctx, device_id):
        super(ResidualBlockV2, self).__init__()
        self.pad = PseudoPadV2(2, npart, ctx, device=device_id)
        self.conv1 = _conv(channels, channels, 3)
        self.relu1 = nn.PReLU(channels)
        self.conv2 = _conv(channels, channels, 3)
        self.relu2 = nn.PReLU(channels)
        self.trim = PseudoFillV2(0, npart, ctx, device=device_id)
        self.__dict__["ctx"] = ctx

    def forward(self, x):
        live = (self.ctx, x.shape[3])
        y = self.conv1.fuse(self.pad(x), self.relu1, live)
        return self.conv2.fuse(y, self.relu2, live, residual=x, trim=self.trim, ring=RING)


class ResidualBlockDown(nn.Module):
    """stride-2 3x3 + 3x3/GDN branch, stride-2 1x1 shortcut (reference: model_zoo_v2.py:95-114)."""

    def __init__(self, channels, channel_in, npart, ctx, device_id):
        super(ResidualBlockDown, self).__init__()
        self.pad1 = PseudoPadV2(1, npart, ctx, device=device_id)
        self.conv1 = _conv(channel_in, channels, 3, 2)
        self.relu1 = nn.PReLU(channels)
        self.pad2 = PseudoPadV2(1, npart, ctx, device=device_id)
        self.conv2 = _conv(channels, channels, 3)
        self.relu2 = PseudoGDNV2(channels, npart, ctx, device_id)
        self.short_cut = _conv(channel_in, channels, 1, 2)
        self.trim = PseudoFillV2(0, npart, ctx, device=device_id)
        self.__dict__["ctx"] = ctx

    def forward(self, x):
        live = (self.ctx, x.shape[3] // 2)
        t = self.short_cut(x, live)
        y = self.conv1.fuse(self.pad1(x), self.relu1, live, ring=RING)
        return self.relu2(self.conv2(self.pad2(y), live), residual=t, trim=self.trim, ring=RING)


class SphereConv2(nn.Module):
    """padded stride-2 3x3 conv (reference: model_zoo_v2.py:116-126)."""

    def __init__(self, channel_in, channel_out, npart, ctx, device_id=0):
        super(SphereConv2, self).__init__()
        self.conv = _conv(channel_in, channel_out, 3, 2)
        self.pad = PseudoPadV2(1, npart, ctx, device=device_id)
        self.trim = PseudoFillV2(0, npart, ctx, device=device_id)
        self.__dict__["ctx"] = ctx

    def forward(self, x):
        return self.conv(self.pad(x), (self.ctx, x.shape[3] // 2), trim=self.trim, ring=RING)


class EncoderV2(nn.Module):
    """analysis transform, 16x down-sampling, sigmoid codes (reference: model_zoo_v2.py:129-151)."""

    def __init__(self, channels, code_channels, npart, ctx, device_id):
        super(EncoderV2, self).__init__()
        down = lambda cin: ResidualBlockDown(channels, cin, npart, ctx, device_id)
        res = lambda: ResidualBlockV2(channels, npart, ctx, device_id)
        att = lambda: AttentionBlock(channels, npart, ctx, device_id)
        self.net = nn.Sequential(
            down(3), res(), down(channels), att(), res(), down(channels), res(),
            SphereConv2(channels, channels, npart, ctx, device_id), att(),
            _conv(channels, code_channels, 1),
        )
        self.act = nn.Sigmoid()
        self.trim = PseudoFillV2(0, npart, ctx, device=device_id)
        self.__dict__["ctx"] = ctx
        backend.watch_state_dict(self)  # packed conv slabs / GDN parameters follow a reload

    def forward(self, x):
        for m in self.net[:-1]:
            x = m(x)
        return self.net[-1](x, (self.ctx, x.shape[3]), sigmoid=True, trim=self.trim)  # self.act fused


class ResidualBlockUp(nn.Module):
    """3x3 -> depth-to-width x2 -> 3x3/IGDN, 1x1 shortcut (reference: model_zoo_v2.py:153-175)."""

    def __init__(self, channels, npart, ctx, device_id):
        super(ResidualBlockUp, self).__init__()
        self.pad1 = PseudoPadV2(1, npart, ctx, device=device_id)
        self.conv1 = _conv(channels, channels * 4, 3)
        self.relu1 = nn.PReLU(channels * 4)
        self.dtow1 = Dtow(2, True, device_id)
        self.pad2 = PseudoPadV2(1, npart, ctx, device=device_id)
        self.conv2 = _conv(channels, channels, 3)
        self.relu2 = PseudoGDNV2(channels, npart, ctx, device_id, inverse=True)
        self.short_cut = _conv(channels, channels * 4, 1)
        self.dtow2 = Dtow(2, True, device_id)
        self.trim = PseudoFillV2(0, npart, ctx, device=device_id)
        self.__dict__["ctx"] = ctx

    def forward(self, x):
        w = x.shape[3]
        br1 = self.conv1.fuse(self.pad1(x), self.relu1, (self.ctx, w), d2w=self.dtow1, ring=RING)
        br2 = self.short_cut(x, (self.ctx, w), d2w=self.dtow2)
        return self.relu2(self.conv2(self.pad2(br1), (self.ctx, 2 * w)), residual=br2, trim=self.trim, ring=RING)


class SphereConvOld(nn.Module):
    """1x1 conv + trim (reference: model_zoo_v2.py:177-186)."""

    def __init__(self, npart, channel_in, channel_out, ctx, device_id=0):
        super(SphereConvOld, self).__init__()
        self.conv = _conv(channel_in, channel_out, 1)
        self.trim = PseudoFillV2(0, npart, ctx, device=device_id)
        self.__dict__["ctx"] = ctx

    def forward(self, x):
        return self.conv(x, (self.ctx, x.shape[3]), trim=self.trim, ring=RING)


class DecoderV2(nn.Module):
    """synthesis transform, 16x up-sampling (reference: model_zoo_v2.py:189-211)."""

    def __init__(self, channels, code_channels, npart, ctx, device_id):
        super(DecoderV2, self).__init__()
        up = lambda: ResidualBlockUp(channels, npart, ctx, device_id)
        res = lambda: ResidualBlockV2(channels, npart, ctx, device_id)
        att = lambda: AttentionBlock(channels, npart, ctx, device_id)
        self.net = nn.Sequential(
            SphereConvOld(npart, code_channels, channels, ctx, device_id),
            att(), res(), up(), res(), up(), att(), res(), up(), res(),
            PseudoPadV2(1, npart, ctx, device=device_id),
            _conv(channels, 12, 3),
            Dtow(2, True, device_id),
        )
        self.__dict__["ctx"] = ctx
        backend.watch_state_dict(self)

    def forward(self, x):
        mods = list(self.net)
        for m in mods[:-2]:  # the blocks and the last pad
            x = m(x)
        # 3x3 conv to 12 channels + the depth-to-width that makes them 3 at full size
        return mods[-2](x, (self.ctx, x.shape[3] - 2), d2w=mods[-1])
