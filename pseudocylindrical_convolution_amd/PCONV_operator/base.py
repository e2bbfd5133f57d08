"""Latitude-tile width weights (reference: PCONV_operator/base.py:5-35).

`set_weight(npart, opt)` returns, per latitude tile, its width in 1/64 units of the
full ERP width.  opt=False: ceil(64 cos(latitude of the tile centre)); opt=True: a
32-band table (optionally read from ./config/param.txt) resampled to npart bands
with a monotone cubic (PCHIP) per hemisphere and rounded up.
"""
import math
import os

import numpy as np
from scipy.interpolate import PchipInterpolator

DEFAULT_BANDS = (8, 18, 24, 36, 46, 58, 62, 62, 62, 62, 63, 63, 63, 63, 63, 63,
                 63, 63, 63, 63, 63, 63, 62, 62, 62, 62, 58, 46, 36, 24, 18, 8)


def load_param(file_name):
    """One comma-separated line of 32 integers, else the built-in table."""
    if os.path.exists(file_name):
        with open(file_name) as f:
            return [int(tok) for tok in f.readline()[:-1].split(',')]
    return list(DEFAULT_BANDS)


def _centre_cos(n):
    # cosine of the latitude of each band centre, north to south
    return np.cos((0.5 - (np.arange(float(n)) + 0.5) / n) * np.pi)


def _merge_pairs(values):
    return [max(values[2 * i], values[2 * i + 1]) for i in range(len(values) // 2)]


def set_weight(npart, opt=False, merge=False, config_file='./config/param.txt'):
    assert npart % 2 == 0, 'npart should be the multiplier of 2 for the merge case'
    bands = npart * 2 if merge else npart
    target = _centre_cos(bands)
    if opt:
        table = np.array([v + 1 for v in load_param(config_file)], dtype=np.float64)
        knots = _centre_cos(32)
        half = bands // 2
        north = PchipInterpolator(knots[:16], table[:16])(target[:half])
        south = PchipInterpolator(knots[16:][::-1], table[16:][::-1])(target[half:])
        out = np.ceil(north).tolist() + np.ceil(south).tolist()
    else:
        out = np.ceil(target * 64.).tolist()
    return _merge_pairs(out) if merge else out
