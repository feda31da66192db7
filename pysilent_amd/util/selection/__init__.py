"""Border mask, per-level top-percent threshold, per-region keypoint indices.

Drop-ins for
  ``pad_inwards``               slam_recognition/util/selection/isolate_rectangle.py:19-23
  ``top_value_points``          slam_recognition/util/selection/top_value_points.py:8-29
  ``max_value_indices_region``  slam_recognition/util/selection/top_value_points.py:32-45
(``isolate_rectangle`` and ``top_value_points_region`` are unused / non-functional in the reference:
SURVEY.md section 2, out of scope.)
"""
import numpy as np

from ... import _runtime
from ..get_dimensions import get_dimensions


def pad_inwards(tensor, paddings):
    """paddings: [[0,0],[top,bottom],[left,right],[0,0]] as the reference passes to tf.pad."""
    get_dimensions(tensor)
    p = [[int(a), int(b)] for a, b in paddings]
    if len(p) != 4 or p[0] != [0, 0] or p[3] != [0, 0]:
        raise ValueError("pad_inwards: only spatial paddings [[0,0],[t,b],[l,r],[0,0]] are supported")
    return _runtime.pad_inwards(tensor, p[1][0], p[1][1], p[2][0], p[2][1])


def top_value_points(color_tensor, top_percent=0.1, value_tensor=None):
    """Zero everything whose value is below (1-p)*max + p*min of its level (batch item)."""
    get_dimensions(color_tensor)
    return _runtime.top_value_points(color_tensor, top_percent, value_tensor)


def _regions_for(op_extents, region_shape):
    if isinstance(region_shape, (list, tuple)) and len(region_shape) and isinstance(region_shape[0], (list, tuple)):
        return [(int(r[0]), int(r[1])) for r in region_shape]            # one (rH, rW) per level
    rh, rw = int(region_shape[1]), int(region_shape[2])                   # the reference's [1, rH, rW, C]
    return [(rh, rw)] * len(op_extents)


def max_value_indices_region(color_tensor, region_shape, value_tensor=None):
    """int64 [K, 4] rows (n, y, x, 0), row-major sorted like tf.where.

    For a rank-4 tensor n is the batch index, exactly as in the reference.  For a PackedPyramid the result
    is a list (one [K_f, 4] array per frame) whose n column is the LEVEL index."""
    get_dimensions(color_tensor)
    value = value_tensor if value_tensor is not None else _runtime.value_from_color(color_tensor)
    packed = isinstance(value, _runtime.PackedPyramid)
    extents = value.extents if packed else [tuple(value.shape[1:3])]
    idx, counts = _runtime.max_value_indices_region(value, _regions_for(extents, region_shape))
    if _runtime.is_torch_tensor(idx):
        idx, counts = idx.cpu().numpy(), counts.cpu().numpy()
    per_frame = [idx[f, :int(counts[f])].copy() for f in range(idx.shape[0])]
    if packed:
        return per_frame
    for f, rows in enumerate(per_frame):
        rows[:, 0] = f
    return np.concatenate(per_frame, axis=0) if per_frame else np.zeros((0, 4), np.int64)
