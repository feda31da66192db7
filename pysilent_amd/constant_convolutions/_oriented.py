"""Shared tail of the oriented generators: scalar profile -> [k..., C_in, C_out] kernel.

Both ``stripe_tensor`` (reference stripe_tensor.py:62-68) and ``end_tensor``
(oriented_end_detector.py:47-53) normalise a scalar profile z in place to sum+ = 1,
sum- = -1 and then expand it as
    K[t, i, o] = |z[t]| * center_in[i] * center_out[o]     where z[t] is "centre"
               = |z[t]| * surround_in[i] * surround_out[o] where z[t] < 0
The two differ only in whether z == 0 counts as centre (it contributes 0 either way).
The reference declares the slot as [len(center_out), len(center_in)] but fills it with an
[in][out] nested list, which only fits for square channel counts: a non-square request
raises ValueError there, and does here too.
"""
import numpy as np


def expand_profile(z, center_in, center_out, surround_in, surround_out):
    ci = np.asarray(center_in, dtype=np.float64)
    co = np.asarray(center_out, dtype=np.float64)
    si = np.asarray(surround_in, dtype=np.float64)
    so = np.asarray(surround_out, dtype=np.float64)
    if len(ci) != len(co) or len(si) != len(so) or len(ci) != len(si):
        raise ValueError("could not broadcast channel lists: oriented kernels need square channel counts")
    a = np.abs(z)[..., None, None]
    # association order of the reference: (out * in) * |z|
    cen = np.multiply.outer(ci, co) * a
    sur = np.multiply.outer(si, so) * a
    neg = (z < 0)[..., None, None]
    return np.where(neg, sur, cen)
