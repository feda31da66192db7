"""Pixel-index tensor.  Mirror of slam_recognition/util/index_tensor.py:7-20 (host constant, NumPy int32):
``from_shape([N, h, w, C])`` -> [h, w, 2]; with ``are_dimensions_reversed`` (the reference's default) channel 0
is x and channel 1 is y -- pinned by the reference's tests/test_index_tensor.py:10-11."""
import numpy as np

are_dimensions_reversed = True


def from_shape(shape):
    dims = [int(d) for d in list(shape)[1:-1]]
    grids = np.meshgrid(*[np.arange(d) for d in dims], indexing="ij")
    if are_dimensions_reversed:
        grids = list(reversed(grids))
    return np.stack(grids, axis=-1).astype(np.int32)


def from_tensor(tensor):
    return from_shape(tensor.shape)
