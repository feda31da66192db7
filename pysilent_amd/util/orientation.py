"""Regular-simplex unit vectors (host side, float64).

Mirror of ``slam_recognition/util/orientation/simplex_coordinates.py:4-34``: the n+1
vertices of a regular simplex centred on the origin of n-space, first vertex on the
first axis.  Pinned by the reference's ``tests/test_simplex_coordinates.py:9-22``.
"""
import numpy as np

__all__ = ["simplex_coordinates", "axis_coordinates", "above_axis_simplex_coordinates"]


def simplex_coordinates(n):
    v = np.zeros((n + 1, n), dtype=np.float64)
    for d in range(n):
        # vertex d closes the unit length with its own axis ...
        v[d, d] = np.sqrt(1.0 - np.dot(v[d, :d], v[d, :d]))
        # ... and every later vertex must make dot(v_d, v_j) = -1/n
        for j in range(d + 1, n + 1):
            v[j, d] = (-1.0 / n - np.dot(v[d, :d], v[j, :d])) / v[d, d]
    return v


def axis_coordinates(n):
    return np.eye(n, n)


def above_axis_simplex_coordinates(n, axis=0):
    s = simplex_coordinates(n)
    s[:, axis] = np.abs(s[:, axis])
    return s
