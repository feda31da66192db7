"""Oriented *stripe* (thin edge) constant kernels (host side, float64, layout [3,3,C_in,C_out]).

Behavioural mirror of ``stripe_tensor`` / ``simplex_stripe_tensors`` /
``rgb_2d_stripe_tensors`` in
slam_recognition/constant_convolutions/edge_orientation_detector/stripe_tensor.py:21-108.
The 7x7 ``edge_tensor`` family of the same reference directory is in ``.edge_tensor``.
"""
import numpy as np

from ...util.attractor import euclidian_attractor_function_generator
from ...util.normalize import normalize_tensor_positive_negative
from ...util.orientation import above_axis_simplex_coordinates
from .._oriented import expand_profile

__all__ = ["stripe_tensor", "simplex_stripe_tensors", "rgb_2d_stripe_tensors"]


def stripe_tensor(normal_vector, center_in, center_out, surround_in, surround_out,
                  attractor_function=euclidian_attractor_function_generator):
    """Profile z(t) = attractor(|((t-1).n) n|): distance of the tap from the facet through the centre."""
    normal = np.asarray(normal_vector, dtype=np.float64)
    ndim = len(normal)
    assert ndim >= 1
    f = attractor_function(ndim)
    z = np.empty((3,) * ndim, dtype=np.float64)
    for t in np.ndindex(*z.shape):
        # sequential sums in the reference's order so the float64 result is identical
        proj = 0
        for ti, ni in zip(t, normal):
            proj = proj + (ti - 1) * ni
        vec = normal * proj
        sq = 0
        for p in vec:
            sq = sq + p ** 2
        z[t] = f(float(np.sqrt(sq)))
    normalize_tensor_positive_negative(z)
    return expand_profile(z, center_in, center_out, surround_in, surround_out)


def simplex_stripe_tensors(dimensions, centers_in, centers_out, surrounds_in, surrounds_out,
                           attractor_function=euclidian_attractor_function_generator):
    normals = above_axis_simplex_coordinates(dimensions)
    return [stripe_tensor(v, ci, co, si, so, attractor_function)
            for v, ci, co, si, so in zip(normals, centers_in, centers_out, surrounds_in, surrounds_out)]


def rgb_2d_stripe_tensors(in_channel=(1, 1, 1)):
    """Three simplex orientations -> R, G, B; every orientation reads the channel SUM.

    The lone .25 in the second surround row is the reference's own asymmetry
    (stripe_tensor.py:108) and is reproduced on purpose.
    """
    x = 2
    ins = [in_channel] * 3
    cen = [[2 * x, -.5 * x, -.5 * x], [-.5 * x, 2 * x, -.5 * x], [-.5 * x, -.5 * x, 2 * x]]
    sur = [[-2 * x, .5 * x, .5 * x], [.25 * x, -2 * x, .5 * x], [.5 * x, .5 * x, -2 * x]]
    return sum(simplex_stripe_tensors(2, ins, cen, ins, sur))
