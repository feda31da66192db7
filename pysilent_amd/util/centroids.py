"""``get_centroids``: value-weighted centroid of every region cell and each pixel's L1 distance to it.

Drop-in for slam_recognition/util/centroids.py:21-46 (the next op of the reference graph after
get_value_from_color, recognition_testing.py:79-84).  ``region_shape`` is the
reference's [1, rh, rw].  Returns (value_centroids like ``value_tensor``, total_pool [N, ceil(h/rh), ceil(w/rw), 1]).
"""
from .. import _runtime
from .get_dimensions import get_dimensions


def get_centroids(value_tensor, region_shape, debug=False):
    get_dimensions(value_tensor)
    return _runtime.centroids(value_tensor, region_shape[1], region_shape[2])
