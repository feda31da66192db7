"""Oriented *line-end* constant kernels (host side, float64, layout [size,size,C_in,C_out]).

Behavioural mirror of ``end_tensor`` / ``simplex_end_tensors`` / ``rgb_2d_end_tensors`` in
slam_recognition/constant_convolutions/oriented_end_detector.py:13-99 (Python-3 semantics:
the tap origin is ``size / 2`` = 1.5 for the default 3x3, i.e. OFF the centre tap, which is
what makes the kernel end- rather than line-selective; SURVEY.md section 7, hard part 8).

``end_bank`` is the build's K-orientation extension used by BASELINE configs 2 and 5
(SURVEY.md section 8d): slice k is the reference's own ``end_tensor`` for
v_k = 3 (cos k pi/K, sin k pi/K); because end_tensor is symmetric under v -> -v the
distinct orientations span pi, not 2 pi.
"""
import math

import numpy as np

from ..util.attractor import linear_attractor_function_generator
from ..util.normalize import normalize_tensor_positive_negative
from ..util.orientation import simplex_coordinates
from ._oriented import expand_profile

__all__ = ["end_tensor", "simplex_end_tensors", "rgb_2d_end_tensors", "end_bank"]


def end_tensor(end_vector, center_in, center_out, surround_in, surround_out,
               attractor_function=linear_attractor_function_generator, size=3):
    """Profile z(t) = attractor(acos(cos angle(t - size/2, v)) - pi/2): +1 across v, 1 - pi along it."""
    v = np.asarray(end_vector, dtype=np.float64)
    ndim = len(v)
    assert ndim >= 1
    f = attractor_function()
    origin = np.asarray([size / 2 for _ in range(ndim)])
    vnorm = np.linalg.norm(v)
    z = np.empty((size,) * ndim, dtype=np.float64)
    for t in np.ndindex(*z.shape):
        r = np.asarray(t) - origin
        c = np.dot(r, v) / (np.linalg.norm(r) * vnorm)
        z[t] = f(((math.acos(c) - math.pi / 2.0) / math.pi) * math.pi)
    normalize_tensor_positive_negative(z)
    return expand_profile(z, center_in, center_out, surround_in, surround_out)


def simplex_end_tensors(dimension, centers_in, centers_out, surrounds_in, surrounds_out,
                        attractor_function=linear_attractor_function_generator, flip=True):
    """One end_tensor per simplex vertex; vertices scaled by 3 and (default) flipped along axis 1."""
    simplex = simplex_coordinates(dimension) * 3
    if flip is not None:
        simplex = np.flip(simplex, int(flip))
    return [end_tensor(v, ci, co, si, so, attractor_function)
            for v, ci, co, si, so in zip(simplex, centers_in, centers_out, surrounds_in, surrounds_out)]


def rgb_2d_end_tensors(north_input_channel=(1, 0, 0), southwest_input_channel=(0, 1, 0),
                       southeast_input_channel=(0, 0, 1)):
    """The reference's 3-orientation simplex bank summed into one 3x3x3x3 kernel."""
    x, xx = 0.5 / 2, -0.25 / 2
    y, yy = 1.0 / 2, 1.0 / 2
    ins = [north_input_channel, southwest_input_channel, southeast_input_channel]
    return sum(simplex_end_tensors(2, ins, [[x, -xx, -xx], [-xx, x, -xx], [-xx, -xx, x]],
                                   ins, [[y, -yy, -yy], [-yy, y, -yy], [-yy, -yy, y]]))


def end_bank(n_orientations, size=3):
    """Grayscale K-orientation bank, HWIO [size,size,1,K] (build-defined extension)."""
    bank = np.empty((size, size, 1, n_orientations), dtype=np.float64)
    for k in range(n_orientations):
        a = k * math.pi / n_orientations
        v = 3.0 * np.array([math.cos(a), math.sin(a)])
        bank[:, :, 0, k] = end_tensor(v, [1], [1], [1], [-1], size=size)[:, :, 0, 0]
    return bank
