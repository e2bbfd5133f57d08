"""Anchor rate-distortion curves the training loop scores a checkpoint against
(reference: test/RDMetric.py:5-15).  The numbers are data: (bpp, viewport MSE on the 0..255
scale) and (bpp, viewport SSIM) pairs of the anchor codec, interpolated with PCHIP."""
import scipy.interpolate

_MSE_RATE = [0.167, 0.1988, 0.2766, 0.315, 0.3714, 0.44, 0.5088, 0.5593, 0.6655, 0.8036, 1.5, 2.3]
_MSE_VALUE = [110.9652, 102.2772, 80.3709, 73.0673, 63.4319, 53.8391, 44.5096, 41.4778, 33.8455, 29.4989, 20, 12]
_SSIM_RATE = [1.553000e-01, 2.204000e-01, 2.670000e-01, 3.438000e-01, 4.372000e-01, 5.103000e-01, 6.798000e-01,
              7.357000e-01, 9.456000e-01, 1.050600e+00, 1.6, 2.3]
_SSIM_VALUE = [8.417000e-01, 8.680000e-01, 8.806000e-01, 8.985000e-01, 9.136000e-01, 9.254000e-01, 9.421000e-01,
               9.456000e-01, 9.592000e-01, 9.640000e-01, 0.978, 0.982]


def mse_tb(x_rt):
    """anchor viewport MSE (images in [0, 1]) at rate x_rt bpp"""
    return scipy.interpolate.pchip_interpolate(_MSE_RATE, _MSE_VALUE, x_rt) / 255 / 255


def ssim_tb(x_rt):
    """anchor viewport SSIM at rate x_rt bpp"""
    return scipy.interpolate.pchip_interpolate(_SSIM_RATE, _SSIM_VALUE, x_rt)
