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

from .PCONV_operator import (ContextReshape, DropGrad, Dtow, EntropyGmm, Extract, MaskConv2, PseudoContextV2,
                             PseudoEntropyContext, PseudoEntropyPad, PseudoFillV2, PseudoGDNV2, PseudoPadV2,
                             PseudoQUANTV2, SphereSlice, SphereUslice, StubMask, backend)

__all__ = ["ClipData", "TileConv2d", "ResidualBlock", "AttentionBlock", "ResidualBlockV2", "ResidualBlockDown",
           "SphereConv2", "EncoderV2", "ResidualBlockUp", "SphereConvOld", "DecoderV2", "EntropyConv",
           "EntropyResidualBlock", "EntropySubNet", "EntropyNet", "CMPNetV2MF", "CMPNetV2M", "CMPNetV2MFExtractor",
           "CMPNetV2Decoder", "CMPNetV2MFEntropy", "AccGrad"]


class _LeakyClip(torch.autograd.Function):
    # identity on [0, 1], slope 0.01 outside (reference: model_zoo_v2.py:8-26)

    @staticmethod
    def forward(ctx, x):
        ops = backend.ops()
        if x.is_cuda and not ctx.needs_input_grad[0] and hasattr(ops, "leaky_clip_"):  # (grad mode is off in here)
            # the codec path: one in-place pass of a hand-written kernel over a copy (the same three roundings;
            # the masked assignments below are two nonzero() passes with a host synchronisation each)
            return ops.leaky_clip_(x.detach().clone().contiguous())
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
    never reads them).  Without fusion support (the CPU oracle backend of the tests) and
    under autograd (training: the library convolution with its own backward, as the
    reference's nn.Conv2d) the same operations run one by one, in the same order.  On the
    HIP backend inference never leaves the hand-written kernels: there is no knob that
    routes it to the vendor library."""

    def _native(self, x, prelu=None, live=None, sigmoid=False, gate=None, residual=None, trim=None, ring=0,
                d2w=None):
        ops = backend.ops()
        slope = prelu.weight if prelu is not None else None
        vendor = not hasattr(ops, "tile_conv2d")
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad):
            # training: the convolution and its activation go through autograd's library
            # kernels, as the reference's nn.Conv2d does; the tile ops around them use their own backward
            vendor, live = True, None
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
        return self.forward_range(x, 0, len(self.net))

    def forward_range(self, x, lo, hi):
        """blocks net[lo:hi] (the whole transform: 0, len(net)).  Lets a caller run the full-size
        stages frame by frame and the small-scale tail on several frames at once (engine.CodecEngine)."""
        last = len(self.net) - 1
        for m in self.net[lo:min(hi, last)]:
            x = m(x)
        if lo <= last < hi:  # (a range that starts behind the final conv is empty: it must not run twice)
            x = self.net[-1](x, (self.ctx, x.shape[3]), sigmoid=True, trim=self.trim)  # self.act fused
        return x


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
        return self.forward_range(x, 0, len(self.net))

    def forward_range(self, x, lo, hi):
        """blocks net[lo:hi] (see EncoderV2.forward_range); the final 3x3 conv and its
        depth-to-width (net[-2], net[-1]) always run together"""
        mods = list(self.net)
        body = len(mods) - 2
        for m in mods[lo:min(hi, body)]:  # the blocks and the last pad
            x = m(x)
        if lo <= body < hi:
            # 3x3 conv to 12 channels + the depth-to-width that makes them 3 at full size
            x = mods[-2](x, (self.ctx, x.shape[3] - 2), d2w=mods[-1])
        return x


# --------------------------------------------------------------------------
# training-time networks: the whole-tensor entropy model and the end-to-end codec
# --------------------------------------------------------------------------
class EntropyConv(nn.Module):
    """causal pad -> 5x5 conv masked to the 3-D causal neighbourhood -> PReLU -> trim
    (reference: model_zoo_v2.py:214-228)."""

    def __init__(self, ngroups, cin, cout, hidden, npart, ctx, device_id, act=True):
        super(EntropyConv, self).__init__()
        self.pad = PseudoEntropyPad(2, npart, ctx, device=device_id)
        self.conv = MaskConv2(ngroups, cin, cout, 5, hidden, device_id)
        self.trim = PseudoFillV2(0, npart, ctx, device=device_id)
        self.act = nn.PReLU(ngroups * cout) if act else None

    def forward(self, x):
        y = self.conv(self.pad(x))
        if self.act is not None:
            y = self.act(y)
        return self.trim(y)


class EntropyResidualBlock(nn.Module):
    """(reference: model_zoo_v2.py:230-239)"""

    def __init__(self, ngroups, cpn, npart, ctx, device_id=0):
        super(EntropyResidualBlock, self).__init__()
        self.conv1 = EntropyConv(ngroups, cpn, cpn, True, npart, ctx, device_id, True)
        self.conv2 = EntropyConv(ngroups, cpn, cpn, True, npart, ctx, device_id, True)

    def forward(self, x):
        return self.conv2(self.conv1(x)) + x


class EntropySubNet(nn.Module):
    """one of the three GMM parameter networks; net_type 0 = weights (softmax), 1 = means,
    2 = scales (ReLU, output bias 2) (reference: model_zoo_v2.py:241-270).  Output (N*G*H*W, K)."""

    def __init__(self, ngroups, cpn, npart, num_gaussian, net_type, ctx, device_id):
        super(EntropySubNet, self).__init__()
        block = lambda: EntropyResidualBlock(ngroups, cpn, npart, ctx, device_id)
        self.net = nn.Sequential(
            EntropyConv(ngroups, 1, cpn, False, npart, ctx, device_id),
            block(), block(), block(), block(), block(),
            EntropyConv(ngroups, cpn, num_gaussian, True, npart, ctx, device_id, False),
        )
        self.reshape = ContextReshape(ngroups, device_id)
        self.act = None
        if net_type == 0:
            self.act = nn.Softmax(dim=1)
        elif net_type == 2:
            self.act = nn.ReLU()
            self.net[6].conv.bias.data.fill_(2)

    def forward(self, x):
        y = self.reshape(self.net(x))
        return self.act(y) if self.act is not None else y


class EntropyNet(nn.Module):
    """rate of every code symbol under its causal GMM, and the mask of the symbols that exist
    (reference: model_zoo_v2.py:272-301)."""

    def __init__(self, ngroups, npart, ctx, cpn=3, num_gaussian=3, device_id=0, drop_flag=False):
        super(EntropyNet, self).__init__()
        self.drop = DropGrad(drop_flag)
        self.weight_net = EntropySubNet(ngroups, cpn, npart, num_gaussian, 0, ctx, device_id)
        self.mean_net = EntropySubNet(ngroups, cpn, npart, num_gaussian, 1, ctx, device_id)
        self.delta_net = EntropySubNet(ngroups, cpn, npart, num_gaussian, 2, ctx, device_id)
        self.mask = None
        self.fill = PseudoFillV2(0, npart, ctx, device=device_id)
        self.fill2 = PseudoFillV2(0, npart, ctx, device=device_id)
        self.ent_loss = EntropyGmm(num_gaussian, device=device_id)

    def setup_mask(self, x):
        with torch.no_grad():
            self.mask = self.fill(torch.ones_like(x).detach()).view(-1)

    def forward(self, x):
        self.setup_mask(x)
        x = self.drop(self.fill2(x))
        weight = self.weight_net(x)
        mean = self.mean_net(x)
        delta = self.delta_net(x) + 1e-6
        loss_vec = self.ent_loss(weight, delta, mean, x.view(-1, 1))
        return loss_vec * self.mask, self.mask


class CMPNetV2MF(nn.Module):
    """end-to-end training graph: reconstruction, per-symbol rate, symbol mask
    (reference: model_zoo_v2.py:304-334)."""

    def __init__(self, valid_dim=162, channels=192, code_channels=192, npart=16, quant_levels=8, opt=False,
                 init=False, device_id=0):
        super(CMPNetV2MF, self).__init__()
        self.slice = SphereSlice(npart, pad=0, opt=opt, device=device_id)
        self.uslice = SphereUslice(npart, pad=0, opt=opt, device=device_id)
        self.ctx = PseudoContextV2(npart, opt, device=device_id)
        self.ctx_ent = PseudoEntropyContext(npart, 1, opt, device=device_id)
        self.encoder = EncoderV2(channels, code_channels, npart, self.ctx, device_id)
        self.decoder = DecoderV2(channels, code_channels, npart, self.ctx, device_id)
        self.quant = PseudoQUANTV2(code_channels, quant_levels, npart, self.ctx, top_alpha=0.0001,
                                   device_id=device_id, ntop=2)
        self.vm = StubMask(valid_dim)
        self.ext = Extract(valid_dim)
        self.clip = ClipData()
        self.ent = EntropyNet(valid_dim // 4, npart, self.ctx_ent, 3, 3, device_id, drop_flag=init)
        self.mean_val = (quant_levels - 1) / 2.
        self.dtw = Dtow(2, True, device_id)

    def forward(self, x):
        code = self.encoder(self.slice(x))
        code_f, code_i = self.quant(code)
        code_f = code_f * self.vm(code_f)
        y = self.uslice(self.decoder(code_f))
        symbols = self.dtw(self.ext(code_i)) - self.mean_val
        ent_vec, mask = self.ent(symbols)
        return self.clip(y), ent_vec, mask


class CMPNetV2M(nn.Module):
    """the transforms and the quantiser without the entropy model.  test/trainDDP_Base.py:109 builds
    `model_zoo_v2.CMPNetV2M`, which the reference's model_zoo_v2.py does not define; this is the
    CMPNetV2MF graph up to the reconstruction, same sub-module names (its checkpoints initialise
    CMPNetV2MF through init_with_trained_model)."""

    def __init__(self, valid_dim=162, channels=192, code_channels=192, npart=16, quant_levels=8, opt=False,
                 init=False, device_id=0):
        super(CMPNetV2M, self).__init__()
        self.slice = SphereSlice(npart, pad=0, opt=opt, device=device_id)
        self.uslice = SphereUslice(npart, pad=0, opt=opt, device=device_id)
        self.ctx = PseudoContextV2(npart, opt, device=device_id)
        self.encoder = EncoderV2(channels, code_channels, npart, self.ctx, device_id)
        self.decoder = DecoderV2(channels, code_channels, npart, self.ctx, device_id)
        self.quant = PseudoQUANTV2(code_channels, quant_levels, npart, self.ctx, top_alpha=0.0001,
                                   device_id=device_id, ntop=2)
        self.vm = StubMask(valid_dim)
        self.clip = ClipData()

    def forward(self, x):
        code_f, _ = self.quant(self.encoder(self.slice(x)))
        code_f = code_f * self.vm(code_f)
        return self.clip(self.uslice(self.decoder(code_f)))


class CMPNetV2MFExtractor(nn.Module):
    """image -> the symbol tensor the entropy model sees (reference: model_zoo_v2.py:336-354)."""

    def __init__(self, valid_dim=162, channels=192, code_channels=192, npart=16, quant_levels=8, opt=False,
                 init=False, device_id=0):
        super(CMPNetV2MFExtractor, self).__init__()
        self.slice = SphereSlice(npart, pad=0, opt=opt, device=device_id)
        self.ctx = PseudoContextV2(npart, opt, device=device_id)
        self.encoder = EncoderV2(channels, code_channels, npart, self.ctx, device_id)
        self.quant = PseudoQUANTV2(code_channels, quant_levels, npart, self.ctx, top_alpha=0.0001,
                                   device_id=device_id, ntop=2)
        self.ext = Extract(valid_dim)
        self.mean_val = (quant_levels - 1) / 2.
        self.dtw = Dtow(2, True, device_id)

    def forward(self, x):
        _, code_i = self.quant(self.encoder(self.slice(x)))
        return self.dtw(self.ext(code_i))


class CMPNetV2Decoder(nn.Module):
    """de-quantised code -> image (reference: model_zoo_v2.py:356-368)."""

    def __init__(self, channels=192, code_channels=192, npart=16, opt=False, init=False, device_id=0):
        super(CMPNetV2Decoder, self).__init__()
        self.uslice = SphereUslice(npart, pad=0, opt=opt, device=device_id)
        self.ctx = PseudoContextV2(npart, opt, device=device_id)
        self.decoder = DecoderV2(channels, code_channels, npart, self.ctx, device_id)
        self.clip = ClipData()

    def forward(self, x):
        return self.clip(self.uslice(self.decoder(x)))


class CMPNetV2MFEntropy(nn.Module):
    """the entropy model on its own (reference: model_zoo_v2.py:370-381)."""

    def __init__(self, valid_dim=162, channels=192, code_channels=192, npart=16, quant_levels=8, opt=False,
                 init=False, device_id=0):
        super(CMPNetV2MFEntropy, self).__init__()
        self.ctx = PseudoEntropyContext(npart, 1, opt, device=device_id)
        self.ent = EntropyNet(valid_dim // 4, npart, self.ctx, 3, 3, device_id, drop_flag=init)
        self.mean_val = (quant_levels - 1) / 2.

    def forward(self, x):
        return self.ent(x - self.mean_val)


class AccGrad():
    """gradient accumulator over `acc_batch` steps (reference: model_zoo_v2.py:383-402)."""

    def __init__(self, params):
        self.acc_grad = [torch.zeros_like(p, memory_format=torch.preserve_format) for p in list(params)]
        self.num_param = len(self.acc_grad)

    def zero(self):
        for g in self.acc_grad:
            g.zero_()

    def acc(self, params):
        torch._foreach_add_(self.acc_grad, [p.grad for p in list(params)])

    def copy_back(self, param):
        torch._foreach_add_([p.grad for p in list(param)], self.acc_grad)
        self.zero()
