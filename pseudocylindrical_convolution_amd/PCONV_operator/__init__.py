"""Operator layer of the codec: the nn.Module surface of the reference's
`PCONV_operator` package (reference: PCONV_operator/__init__.py:1-16), over the
MI355X-native `PCONV` / `coder` modules."""
from . import backend
from .BaseOpModule import BaseOpModule
from .GDN import LowerBound
from .base import set_weight
from .pytorch_ssim import SSIM
from .util import Logger, ModuleSaver, Timer
from .ops import (MultiProject, MultiProjectM, Dtow, EntropyGmm, ContextReshape, DropGrad, MaskConv2, MaskConv3,
                  SphereSlice, SphereUslice, StubMask, Extract, EntropyGmmTable, EntropyBatchGmmTable,
                  EntropyContextNew, EntropyConv2, EntropyConv2Batch, EntropyCtxPadRun2, DExtract2, DInput2,
                  DExtract2Batch, EntropyAdd, EntropyConvD, EntropyResidualBlockD, PseudoFillV2,
                  PseudoContextV2, PseudoGDNV2, PseudoPadV2, PseudoEntropyContext, PseudoEntropyPad,
                  PseudoQUANTV2, PseudoDQUANT)
