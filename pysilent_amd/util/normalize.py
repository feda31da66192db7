"""Kernel normaliser (host side, float64).

Mirror of ``slam_recognition/util/normalize/normalize_center_surround.py:5-24``: scale the
strictly positive entries so they sum to ``positive_value`` and the strictly negative
entries so they sum to ``-negative_value``.  Like the reference it works IN PLACE and
returns the same array object (pinned by the reference's
``tests/test_normalize_center_surround.py:24-26``).
"""
import numpy as np

__all__ = ["normalize_tensor_positive_negative"]


def normalize_tensor_positive_negative(tensor, positive_value=1.0, negative_value=1.0, epsilon=1e-12):
    if not isinstance(tensor, np.ndarray):
        raise TypeError("normalize_tensor_positive_negative works in place on a numpy array")
    pos = tensor > 0
    neg = tensor < 0
    # sequential left-to-right sums, as the reference's Python ``sum`` over nditer does
    sum_pos = max(float(sum(tensor[pos].ravel().tolist())), epsilon)
    sum_neg = max(float(sum((-tensor[neg]).ravel().tolist())), epsilon)
    tensor[pos] *= positive_value / sum_pos
    tensor[neg] *= negative_value / sum_neg
    return tensor
