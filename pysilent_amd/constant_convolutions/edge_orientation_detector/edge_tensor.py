"""7x7 thick-edge / boundary kernels (host side, float64, layout [7,7,C_in,C_out]) -- SURVEY.md section 8f rank 4.

Behavioural mirror of ``edge_tensor`` / ``simplex_edge_tensors`` / ``rgb_2d_edge_tensors`` /
``rgb_2d_edge_tensors_time_diff`` / ``rgb_2d_end_tensors`` in
slam_recognition/constant_convolutions/edge_orientation_detector/edge_tensor.py:21-158.  No filter of the
reference uses them; they run through the same 7x7x3x3 HIP stencil as the regulator's blur
(``silent_conv2d_same``).  Reference behaviour kept on purpose: the facet passes through tap (1, 1) of the
7-wide grid (``t - 1``, edge_tensor.py:56), not through its centre (3, 3).
"""
import numpy as np

from ...util.attractor import euclidian_attractor_function_generator
from ...util.normalize import normalize_tensor_positive_negative
from ...util.orientation import simplex_coordinates
from .._oriented import expand_profile

__all__ = ["edge_tensor", "simplex_edge_tensors", "rgb_2d_edge_tensors", "rgb_2d_edge_tensors_time_diff",
           "rgb_2d_end_tensors"]

_SIDE = 7


def edge_tensor(normal_vector, center_in, center_out, surround_in, surround_out,
                attractor_function=euclidian_attractor_function_generator):
    """Profile z(t) = attractor(signed distance of tap t from the facet through tap (1,..,1)); the attractor is
    built with max_positive = 0, max_negative = -1, i.e. 0 on the facet rising towards +1 on the side the normal
    points to and falling towards -1 on the other.  z >= 0 is "centre", z < 0 "surround"."""
    normal = np.asarray(normal_vector, dtype=np.float64)
    ndim = len(normal)
    assert ndim >= 1
    f = attractor_function(ndim, max_positive=0.0, max_negative=-1.0)
    z = np.empty((_SIDE,) * ndim, dtype=np.float64)
    for t in np.ndindex(*z.shape):
        # sequential sums in the reference's order so the float64 result is identical
        proj = 0
        for ti, ni in zip(t, normal):
            proj = proj + (ti - 1) * ni
        signed = 0
        for p in normal * proj:
            signed = signed + p * abs(p)
        dist = float(np.sqrt(abs(signed)))
        z[t] = f(dist if signed >= 0 else -dist)
    normalize_tensor_positive_negative(z)
    return expand_profile(z, center_in, center_out, surround_in, surround_out)


def simplex_edge_tensors(dimensions, centers_in, centers_out, surrounds_in, surrounds_out,
                         attractor_function=euclidian_attractor_function_generator, flip=None):
    """One edge tensor per vertex of the regular simplex (d + 1 orientations, positive responses only);
    ``flip`` reverses the vertex table along that axis first (edge_tensor.py:100-102)."""
    normals = simplex_coordinates(dimensions)
    if flip is not None:
        normals = np.flip(normals, flip)
    return [edge_tensor(v, ci, co, si, so, attractor_function)
            for v, ci, co, si, so in zip(normals, centers_in, centers_out, surrounds_in, surrounds_out)]


def rgb_2d_edge_tensors(in_channel=(1, 1, 1)):
    """Three simplex orientations -> R, G, B, each reading the channel sum (edge_tensor.py:109-121)."""
    x = 2
    ins = [in_channel] * 3
    cen = [[2 * x, -.5 * x, -.5 * x], [-.5 * x, 2 * x, -.5 * x], [-.5 * x, -.5 * x, 2 * x]]
    sur = [[-2 * x, .5 * x, .5 * x], [.5 * x, -2 * x, .5 * x], [.5 * x, .5 * x, -2 * x]]
    return sum(simplex_edge_tensors(2, ins, cen, ins, sur))


def rgb_2d_edge_tensors_time_diff(in_channel=(1, 1, 1), surround_in_channel=(-1, -1, -1)):
    """Same with a separate (negated) surround input and full-strength opponent outputs (edge_tensor.py:124-137)."""
    x = 2
    cen = [[2 * x, -x, -x], [-x, 2 * x, -x], [-x, -x, 2 * x]]
    sur = [[-2 * x, x, x], [x, -2 * x, x], [x, x, -2 * x]]
    return sum(simplex_edge_tensors(2, [in_channel] * 3, cen, [surround_in_channel] * 3, sur))


def rgb_2d_end_tensors(north_input_channel=(1, -.5, -.5), southwest_input_channel=(-.5, 1, -.5),
                       southeast_input_channel=(-.5, -.5, 1)):
    """Legacy 7x7 line-end bank (edge_tensor.py:140-158): every orientation reads the two colours of the OTHER
    two stripe orientations; the three arguments are accepted and ignored, as in the reference."""
    x = 2
    ins = [(0, 1, 1), (1, 0, 1), (1, 1, 0)]
    outs = [[2 * x, -x, -x], [-x, 2 * x, -x], [-x, -x, 2 * x]]
    return sum(simplex_edge_tensors(2, ins, outs, ins, outs, flip=1))
