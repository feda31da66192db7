"""Isotropic "blur" constant kernel used by the divisive regulator (host side, float64).

Behavioural mirror of ``blur_tensor`` in
slam_recognition/constant_convolutions/gaussian_blur/gaussian_blur.py:13-54: every
(in, out) channel pair holds the same scalar profile attractor(r) with max_negative = 0,
i.e. 1 / (2 r + 1) in 2-D -- so the blur acts on the channel SUM.
"""
import numpy as np

from ..util.attractor import euclidian_attractor_function_generator

__all__ = ["blur_tensor", "blur_profile"]


def blur_profile(n, lengths=3, attractor_function=euclidian_attractor_function_generator):
    """The scalar profile shared by every channel pair of ``blur_tensor``."""
    assert n >= 1
    f = attractor_function(n, max_negative=0)
    shape = [lengths] * n if isinstance(lengths, int) else [lengths[i] for i in range(n)]
    prof = np.empty(shape, dtype=np.float64)
    for t in np.ndindex(*shape):
        sq = 0
        for ti, li in zip(t, shape):
            sq = sq + (ti - int(li / 2)) ** 2
        prof[t] = f(float(np.sqrt(sq)))
    return prof


def blur_tensor(n, lengths=3, channels_in=3, channels_out=3,
                attractor_function=euclidian_attractor_function_generator):
    prof = blur_profile(n, lengths, attractor_function)
    return np.broadcast_to(prof[..., None, None], prof.shape + (channels_in, channels_out)).copy()
