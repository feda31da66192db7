"""``apply_filter(tensor, filter)``: stride-1 SAME cross-correlation with a constant HWIO kernel.

Drop-in for slam_recognition/util/apply_filter.py:4-7, but EAGER: runs the gfx950 stencil kernel and
returns float32 data instead of a symbolic tf.Tensor.  ``relu`` / ``clip_hi`` are extensions that fuse
the ``tf.maximum(.., [0])`` / ``tf.clip_by_value(.., 0, hi)`` the reference applies right after
(recognition_testing.py:73-74) into the same launch.
"""
from .. import _runtime
from .get_dimensions import get_dimensions


def apply_filter(tensor, filter, relu=False, clip_hi=None):
    get_dimensions(tensor)   # raises the reference's TypeError for foreign types
    return _runtime.conv2d_same(tensor, filter, relu=relu, clip_hi=clip_hi)
