"""``get_centroids``: value-weighted centroid of every region cell and each pixel's L1 distance to it.

Drop-in for slam_recognition/util/centroids.py:21-46 (the next op of the reference graph after
get_value_from_color, recognition_testing.py:79-84).  ``region_shape`` is the
reference's [1, rh, rw].  Returns (value_centroids like ``value_tensor``, total_pool [N, ceil(h/rh), ceil(w/rw), 1]).
"""
import numpy as np

from .. import _runtime
from .get_dimensions import get_dimensions


def additive_filter(shape, channels):
    """The box-sum kernel of the centroid pools, slam_recognition/util/centroids.py:9-18: [rh, rw, channels, channels] with the
    2 x 2 identity at every tap (host constant, float32).  The reference writes a literal [[1, 0], [0, 1]] into every
    tap, so any channel count but 2 fails to broadcast -- kept.  silent_centroids applies these sums inside its kernels;
    the constant is here for callers that build their own pools."""
    shape = [int(d) for d in shape]
    filter_out = np.zeros(shape + [int(channels), int(channels)])
    filter_out[...] = np.asarray([[1, 0], [0, 1]])          # ValueError unless channels == 2, like the reference
    return filter_out.astype(np.float32)


def get_centroids(value_tensor, region_shape, debug=False):
    get_dimensions(value_tensor)
    return _runtime.centroids(value_tensor, region_shape[1], region_shape[2])
