"""``rgby_filter``: colour-opponent center-surround (V1 blobs) + ReLU.
Drop-in for slam_recognition/filters/rgby.py:6-14."""
from .. import _runtime
from ..constant_convolutions.center_surround import rgby_3
from ..util.get_dimensions import get_dimensions

_cache = {}


def rgby_filter(tensor):
    n = get_dimensions(tensor)
    if n not in _cache:
        _cache[n] = rgby_3(n).reshape(3, 3, 3, 3)
    return _runtime.conv2d_same(tensor, _cache[n], relu=True)
