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
    backend's tile convolution.  `fuse(x, prelu)` folds a following nn.PReLU into
    the kernel epilogue.  `live=(ctx, base)` tells the kernel which output columns
    can ever be read: tile t only needs columns < widths_t(base) + (wo - base),
    base = tile width of the scale the output lives in; 64-column blocks beyond
    that are written as zeros without being computed (they are dead: every
    consumer either trims them or never reads them).
    PCONV_TILE_CONV=vendor routes to torch's own conv for A/B timing on the GPU
    (never used for parity claims)."""

    def _native(self, x, prelu, live=None):
        ops = backend.ops()
        slope = prelu.weight if prelu is not None else None
        if os.environ.get("PCONV_TILE_CONV", "native") == "vendor" or not hasattr(ops, "tile_conv2d"):
            y = nn.functional.conv2d(x, self.weight, self.bias, self.stride)
            return nn.functional.prelu(y, slope) if slope is not None else y
        limit, npart = None, 0
        if live is not None and hasattr(ops, "conv_col_limit") and os.environ.get("PCONV_SKIP_DEAD", "1") == "1":
            ctx, base = live
            k, s = self.kernel_size[0], self.stride[0]
            wo = (x.shape[3] - k) // s + 1
            limit, npart = ops.conv_col_limit(ctx.native(x), x.shape[2], base, wo - base, x)
        return ops.tile_conv2d(self, x, self.weight, self.bias, self.stride[0], slope, limit, npart)

    def forward(self, x, live=None):
        return self._native(x, None, live)

    def fuse(self, x, prelu, live=None):
        return self._native(x, prelu, live)


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
        return self.trim(x + self.conv3(y, live))


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
        a = x
        for i, m in enumerate(self.attention):
            a = m(a, (self.ctx, x.shape[3])) if i == 3 else m(a)
        return self.trim(x + self.trunk(x) * a)


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
        y = self.conv2.fuse(y, self.relu2, live)
        return self.trim(x + y)


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
        y = self.conv1.fuse(self.pad1(x), self.relu1, live)
        y = self.relu2(self.conv2(self.pad2(y), live))
        return self.trim(t + y)


class SphereConv2(nn.Module):
    """padded stride-2 3x3 conv (reference: model_zoo_v2.py:116-126)."""

    def __init__(self, channel_in, channel_out, npart, ctx, device_id=0):
        super(SphereConv2, self).__init__()
        self.conv = _conv(channel_in, channel_out, 3, 2)
        self.pad = PseudoPadV2(1, npart, ctx, device=device_id)
        self.trim = PseudoFillV2(0, npart, ctx, device=device_id)
        self.__dict__["ctx"] = ctx

    def forward(self, x):
        return self.trim(self.conv(self.pad(x), (self.ctx, x.shape[3] // 2)))


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

    def forward(self, x):
        for i, m in enumerate(self.net):
            x = m(x, (self.ctx, x.shape[3])) if i == len(self.net) - 1 else m(x)
        return self.trim(self.act(x))


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
        br1 = self.dtow1(self.conv1.fuse(self.pad1(x), self.relu1, (self.ctx, w)))
        br1 = self.relu2(self.conv2(self.pad2(br1), (self.ctx, 2 * w)))
        br2 = self.dtow2(self.short_cut(x, (self.ctx, w)))
        return self.trim(br1 + br2)


class SphereConvOld(nn.Module):
    """1x1 conv + trim (reference: model_zoo_v2.py:177-186)."""

    def __init__(self, npart, channel_in, channel_out, ctx, device_id=0):
        super(SphereConvOld, self).__init__()
        self.conv = _conv(channel_in, channel_out, 1)
        self.trim = PseudoFillV2(0, npart, ctx, device=device_id)
        self.__dict__["ctx"] = ctx

    def forward(self, x):
        return self.trim(self.conv(x, (self.ctx, x.shape[3])))


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

    def forward(self, x):
        for m in self.net:
            x = m(x, (self.ctx, x.shape[3] - 2)) if isinstance(m, TileConv2d) else m(x)
        return x
