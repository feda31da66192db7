"""``orientation_filter``: oriented stripe detector + ReLU, then the blur-based divisive regulator.
Drop-in for slam_recognition/filters/orientation.py:12-35 (regulation value 1.0, root 0.1 hard-wired
there at :33)."""
from .. import _runtime
from ..constant_convolutions.edge_orientation_detector import rgb_2d_stripe_tensors
from ..constant_convolutions.gaussian_blur import blur_tensor
from ..util.get_dimensions import get_dimensions

_cache = {}


def orientation_filter(tensor, blur_size=7, flat_policy="ieee"):
    dims = get_dimensions(tensor)
    key = (dims, int(blur_size))
    if key not in _cache:
        _cache[key] = (rgb_2d_stripe_tensors().reshape(3, 3, 3, 3),
                       blur_tensor(dims, lengths=blur_size).reshape(blur_size, blur_size, 3, 3))
    stripe, blur = _cache[key]
    orient = _runtime.conv2d_same(tensor, stripe, relu=True)
    return _runtime.regulate(orient, blur, 1.0, .1, flat_policy)
