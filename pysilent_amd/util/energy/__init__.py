"""3x3 non-max suppression (the stateless core of the reference's energy / boosting code).

``local_maxima``  x * where(x == maxpool3x3(x), x, 0)   slam_recognition/_experimental/vision_filter.py:88-89
``has_fired``     where(x == maxpool3x3(x), 1, 0)        slam_recognition/util/energy/boosting.py:18-22
The stateful exhaustion update is in .boosting (get_boosting / initialize_boosting) and .recovery.
"""
from ... import _runtime
from ..get_dimensions import get_dimensions


def local_maxima(tensor):
    get_dimensions(tensor)
    return _runtime.nms3x3(tensor, "product")


def has_fired(tensor):
    get_dimensions(tensor)
    return _runtime.nms3x3(tensor, "fired")


from .boosting import get_boosting, initialize_boosting  # noqa: E402
from .recovery import generate_recovery  # noqa: E402,F401
