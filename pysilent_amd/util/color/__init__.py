"""Colour helpers.  ``get_value_from_color`` mirrors slam_recognition/util/color/get_value.py:6-12:
channel sum times float32(1/C), keepdims; ``get_bw_from_color`` util/color/get_bw.py:6-13; ``to_channels``
util/color/to_channels.py:6-16."""
import numpy as np

from ... import _runtime
from ..get_dimensions import get_dimensions


def get_value_from_color(color_tensor):
    get_dimensions(color_tensor)
    return _runtime.value_from_color(color_tensor)


def get_bw_from_color(color_tensor):
    """Mirror of util/color/get_bw.py:6-13: 1 where the channel sum is not 0, else 0 (one channel)."""
    get_dimensions(color_tensor)
    return _runtime.bw_from_color(color_tensor)


def to_channels(images, num_channels=3, name=None):
    """Repeat a 1-channel map ``num_channels`` times along the last axis (the reference tiles the last axis and then pins
    its extent to ``num_channels``, to_channels.py:13-15, which only a 1-channel input satisfies: anything else raises).
    Runs on the GPU as a 1 x 1 convolution with a kernel of ones (x * 1 is exact)."""
    get_dimensions(images)
    channels = images.channels if isinstance(images, _runtime.PackedPyramid) else int(images.shape[-1])
    if channels != 1:
        raise ValueError("to_channels: last dimension is %d, the tiled shape would be %d, not %d"
                         % (channels, channels * int(num_channels), int(num_channels)))
    if int(num_channels) < 1:
        raise ValueError("to_channels: num_channels must be >= 1")
    return _runtime.conv2d_same(images, np.ones((1, 1, 1, int(num_channels)), np.float32))
