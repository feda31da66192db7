"""Colour helpers.  ``get_value_from_color`` mirrors slam_recognition/util/color/get_value.py:6-12:
channel sum times float32(1/C), keepdims."""
from ... import _runtime
from ..get_dimensions import get_dimensions


def get_value_from_color(color_tensor):
    get_dimensions(color_tensor)
    return _runtime.value_from_color(color_tensor)
