"""``rgc_filter``: per-colour center-surround (retinal ganglion cells) + ReLU.
Drop-in for slam_recognition/filters/rgc.py:6-18."""
from .. import _runtime
from ..constant_convolutions.center_surround import midget_rgc
from ..util.get_dimensions import get_dimensions

_cache = {}


def rgc_filter(tensor):
    n = get_dimensions(tensor)
    if n not in _cache:
        _cache[n] = midget_rgc(n).reshape(3, 3, 3, 3)       # tf.constant(rgc, shape=(3, 3, 3, 3))
    return _runtime.conv2d_same(tensor, _cache[n], relu=True)
