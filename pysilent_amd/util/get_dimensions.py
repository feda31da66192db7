"""Spatial rank of an NHWC tensor.  Mirror of slam_recognition/util/get_dimensions.py:7-15 (same
TypeError text for anything that is not a tensor or ndarray)."""
import numpy as np

from .._runtime import PackedPyramid, TYPE_ERROR_MESSAGE, is_torch_tensor


def get_dimensions(tensor):
    if isinstance(tensor, np.ndarray) or is_torch_tensor(tensor):
        return len(tensor.shape) - 2
    if isinstance(tensor, PackedPyramid):
        return 2
    raise TypeError(TYPE_ERROR_MESSAGE)
