"""Distance -> weight curves used by the constant-kernel generators (host side, float64).

Behavioural mirror of the reference's
``slam_recognition/util/attractor/euclidian_attractor_function.py:8-30`` and
``slam_recognition/util/attractor/linear_attractor_function.py:8-28``.  Both factories
return a callable; the callables here also accept NumPy arrays (the reference's only
take scalars), which is what lets the generators in ``constant_convolutions`` be written
without per-tap Python loops.
"""
import numpy as np

__all__ = ["euclidian_attractor_function_generator", "linear_attractor_function_generator"]


def euclidian_attractor_function_generator(n, max_positive=1.0, max_negative=1.0):
    """f(x) = (p + n) / (2 x^(d-1) + 1)^(d-1) - n for x >= 0, odd-extended for x < 0.

    ``n`` is the number of spatial dimensions d (the reference's argument name).
    """
    span = float(max_positive) + float(max_negative)
    e = n - 1

    def euclid(x):
        x = np.asarray(x, dtype=np.float64)
        ax = np.abs(x)
        core = span / ((2.0 * ax ** e + 1.0) ** e) - max_negative
        out = np.where(x >= 0, core, -core)
        return out if out.ndim else float(out)

    return euclid


def linear_attractor_function_generator(max_positive=1.0, max_negative=1.0):
    """f(x) = p - (n + p) |x|: a tent through (0, p) and (+-1, -n)."""
    slope = float(max_negative) + float(max_positive)

    def tent(x):
        x = np.asarray(x, dtype=np.float64)
        out = max_positive - slope * np.abs(x)
        return out if out.ndim else float(out)

    return tent
