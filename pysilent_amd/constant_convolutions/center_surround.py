"""Center-surround ("top-hat") constant kernels (host side, float64, layout [3]*ndim + [C_in, C_out]).

Behavioural mirror of the reference generators
  * ``center_surround_tensor``  slam_recognition/constant_convolutions/center_surround/center_surround_tensor.py:17-48
  * ``midget_rgc`` / ``midget_rgc_full``  .../center_surround/rgc.py:14-54
  * ``rgby`` / ``rgby_3``  .../center_surround/rgby.py:14-56
pinned by the reference's tests/test_center_surround_tensors.py:8-63 and by
tests/golden/kernels.npz (generated from the reference itself).

Structure exploited here (and by the HIP kernels): the kernel is ONE scalar 3^n profile
``1/sqrt(manhattan distance)`` times a channel-mix matrix for the surround taps, plus a
second channel-mix matrix on the centre tap scaled by the sum of the surround profile.
"""
import numpy as np

from ..util.normalize import normalize_tensor_positive_negative

__all__ = ["center_surround_tensor", "midget_rgc", "midget_rgc_full", "rgby", "rgby_3"]


def _surround_profile(ndim):
    """3^ndim array of 1/sqrt(L1 distance to the centre); 0 at the centre; and its sum."""
    axes = np.indices((3,) * ndim)
    manhattan = np.abs(axes - 1).sum(axis=0)
    prof = np.zeros((3,) * ndim, dtype=np.float64)
    off = manhattan > 0
    prof[off] = 1.0 / np.sqrt(manhattan[off])
    # the reference accumulates ``total`` tap by tap in C order, starting from int 0
    total = 0
    for w in prof.ravel().tolist():
        if w != 0.0:
            total += w
    return prof, total


def center_surround_tensor(ndim, center_in, center_out, surround_in, surround_out):
    """K[t, i, o] = surround_out[o]*surround_in[i]/sqrt(|t-1|_1) off centre,
    K[1.., i, o] = center_out[o]*center_in[i]*sum(surround profile) at the centre."""
    assert ndim >= 1
    prof, total = _surround_profile(ndim)
    s_mix = np.multiply.outer(np.asarray(surround_in, dtype=np.float64),
                              np.asarray(surround_out, dtype=np.float64))
    c_mix = np.multiply.outer(np.asarray(center_in, dtype=np.float64),
                              np.asarray(center_out, dtype=np.float64))
    if s_mix.shape != (len(center_in), len(center_out)):
        raise ValueError("surround channel lists must match the centre channel lists in length")
    # (o*i)*w in the reference's association order
    k = s_mix[(None,) * ndim] * prof[(...,) + (None, None)]
    k[(1,) * ndim] = c_mix * total
    return k


def _sum_of(ndim, rows):
    acc = None
    for ci, co, si, so in rows:
        t = center_surround_tensor(ndim, center_in=ci, center_out=co, surround_in=si, surround_out=so)
        acc = t if acc is None else acc + t
    return acc


def midget_rgc(n):
    """Per-colour center-surround (retinal ganglion): channel-diagonal; normalised to sum+ = 4, sum- = -2."""
    d = 1.0
    e = np.eye(3) * d
    rows = [(list(e[c]), list(e[c]), list(e[c]), list(-e[c])) for c in range(3)]
    return normalize_tensor_positive_negative(_sum_of(n, rows), 4.0, 2.0)


def midget_rgc_full(n):
    d = 0.5
    e = np.eye(3) * d
    rows = [(list(e[c]), list(-e[c]), list(e[c]), list(e[c])) for c in range(3)]
    return normalize_tensor_positive_negative(_sum_of(n, rows), 4.0, 2.0)


def rgby(n):
    """Colour-opponent center-surround with a 4th (yellow) output channel; NOT normalised."""
    d = 1.0
    rows = [
        ([0, 0, d], [0, 0, d, 0], [0, d, 0], [0, 0, -d, 0]),                 # red | green
        ([d, 0, 0], [d, 0, 0, 0], [0, d / 2, d / 2], [-d, 0, 0, 0]),         # blue | yellow
        ([0, d, 0], [0, d, 0, 0], [0, 0, d], [0, -d, 0, 0]),                 # green | red
        ([0, d / 2, d / 2], [0, 0, 0, d], [d, 0, 0], [0, 0, 0, -d]),         # yellow | blue
    ]
    return _sum_of(n, rows)


def rgby_3(n):
    """Three-output colour-opponent center-surround, normalised to sum+ = 4, sum- = -2."""
    d = 1.0 / 3
    rows = [
        ([0, 0, d], [0, 0, d], [0, d, 0], [0, 0, -d]),
        ([d, 0, 0], [d, 0, 0], [0, d / 2, d / 2], [-d, 0, 0]),
        ([0, d, 0], [0, d, 0], [0, 0, d], [0, -d, 0]),
        ([0, d / 2, d / 2], [0, d / 2, d / 2], [d, 0, 0], [0, -d / 2, -d / 2]),
    ]
    return normalize_tensor_positive_negative(_sum_of(n, rows), 4.0, 2.0)
