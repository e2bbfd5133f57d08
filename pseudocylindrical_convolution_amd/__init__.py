"""pseudocylindrical_convolution_amd -- MI355X-native encode/decode path of the
pseudocylindrical-convolution 360-degree image codec.

Layout
  csrc/            hand-written HIP kernels (gfx950) + host geometry + CPU coder
  PCONV.py         drop-in for the reference's native module `PCONV`
  coder.py         drop-in for the reference's native module `coder`
  PCONV_operator/  the nn.Module operator surface
  model_zoo_v2.py  analysis / synthesis transforms
  pseudo_codec.py  entropy coder loops, PseudoEncoder / PseudoDecoder, CLI
"""
import sys as _sys

__version__ = "0.1.0"


def install_dropin():
    """Register this package's modules under the reference's import names
    (`PCONV`, `coder`, `PCONV_operator`), so that code written against the
    reference (its model_zoo_v2.py / pseudo_codec.py) imports them unchanged."""
    from . import PCONV as _ops, coder as _coder, PCONV_operator as _operator
    _sys.modules.setdefault("PCONV", _ops)
    _sys.modules.setdefault("coder", _coder)
    _sys.modules.setdefault("PCONV_operator", _operator)
    for name, mod in list(_sys.modules.items()):
        if name.startswith(__name__ + ".PCONV_operator."):
            _sys.modules.setdefault("PCONV_operator." + name.rsplit(".", 1)[1], mod)
    return _ops, _coder, _operator
