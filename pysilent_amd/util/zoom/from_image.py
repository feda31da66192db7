"""``image_to_zoom_tensor``: image -> fixed-size zoom pyramid [L, h, w, C], on the GPU.

Drop-in for slam_recognition/util/zoom/from_image.py:10-69.  The host part below reproduces the
reference's (and scipy.ndimage.zoom's) *integer* decisions -- number of levels, crop slices, Python
``round`` of the zoomed extents -- and hands explicit geometry to the HIP resampler, which implements
scipy's un-prefiltered order-5 spline (float64 tap tables built in the library).  Differences, both
documented in SURVEY.md section 8a-1: canvas pixels the zoomed crop does not cover are 0 here
(uninitialised memory in the reference), and the result is float32 (the reference returns a float64
array holding float32 values).
"""
import math

import numpy as np

from ... import _runtime

_plan_cache = {}


def reference_levels(image_hw, center_dimensions, scale):
    """Level geometry of image_to_zoom_tensor: list of (y0, x0, crop_h, crop_w, zoom_h, zoom_w, out_h, out_w).

    ``center_dimensions`` is (w, h) like the reference's argument (it reverses it, from_image.py:44)."""
    center_hw = list(reversed([int(d) for d in center_dimensions]))
    num_scales = int(math.ceil(max(math.log(i / c, scale) for i, c in zip(image_hw, center_hw))))
    levels = []
    for s in range(num_scales):
        geo = []
        for i, c in zip(image_hw, center_hw):
            span = c * (scale ** s)
            lo = int(max((i - span) / 2, 0))
            hi = min(int((i + span) / 2), i)
            geo.append((lo, hi - lo))
        z = 1.0 / (scale ** s)
        (y0, ch), (x0, cw) = geo
        levels.append((y0, x0, ch, cw, int(round(ch * z)), int(round(cw * z)), center_hw[0], center_hw[1]))
    return levels


def classic_levels(image_hw, scale, n_levels):
    """'Classic layout' (SURVEY.md section 8d): level l = the WHOLE frame resampled by scale**-l, stored
    compactly at its own extents (what from_image.py:49-64 computes when the crop clips to the frame)."""
    h, w = image_hw
    levels = []
    for l in range(int(n_levels)):
        z = 1.0 / (scale ** l)
        zh, zw = int(round(h * z)), int(round(w * z))
        if zh < 1 or zw < 1:
            raise ValueError("level %d of a %dx%d frame at scale %g is empty" % (l, h, w, scale))
        levels.append((0, 0, h, w, zh, zw, zh, zw))
    return levels


def _plan(frame_shape, levels, device):
    key = (tuple(frame_shape), tuple(levels), device)
    plan = _plan_cache.get(key)
    if plan is None:
        if len(_plan_cache) > 32:
            _plan_cache.clear()
        plan = _plan_cache[key] = _runtime.PyramidPlan(frame_shape[0], frame_shape[1], frame_shape[2], levels, device)
    return plan


def _frames(image, num_colors=None):
    if not (isinstance(image, np.ndarray) or _runtime.is_torch_tensor(image)):
        raise TypeError(_runtime.TYPE_ERROR_MESSAGE)
    if image.ndim == 3:
        image = image[None]
    if image.ndim != 4:
        raise ValueError("image must be [H, W, C] or [n, H, W, C]")
    if num_colors is not None and image.shape[-1] != num_colors:
        raise ValueError("image has %d colours, num_colors says %d" % (image.shape[-1], num_colors))
    dev = image.device.index or 0 if _runtime.is_torch_tensor(image) else None
    return image, dev


def image_to_zoom_tensor(image, num_colors, center_dimensions, scale):
    """[H, W, C] image -> float32 [L, h, w, C] pyramid (same arguments as the reference)."""
    assert scale > 1, "Scale must be greater than one."
    assert num_colors > 0, "Number of colors must be greater than zero."
    for d in center_dimensions:
        assert d > 0, "Each dimension must be larger than zero."
    frames, dev = _frames(image, num_colors)
    if frames.shape[0] != 1:
        raise ValueError("from_image takes one [H, W, C] image; use classic_pyramid / PyramidPlan for batches")
    if int(num_colors) not in (1, 3):
        # the kernels take 1 or 3 interleaved channels; the reference zooms every colour plane on its own
        # (from_image.py:54-64), so any other count is that many single-channel pyramids side by side
        n_col = int(num_colors)
        if _runtime.is_torch_tensor(frames):
            import torch
            src = frames[0]
            if not src.is_contiguous():
                raise ValueError("GPU tensors must be contiguous (NHWC, C innermost)")
            npx = int(src.shape[0] * src.shape[1])
            out = None
            for c in range(n_col):
                # plane c cut out of the interleaved image (and widened) by the library's strided cast ...
                plane = torch.empty((src.shape[0], src.shape[1], 1), dtype=torch.float32, device=src.device)
                _runtime.cast_interleave(src, plane, n_col, c, 1, 1, 0, npx)
                z = image_to_zoom_tensor(plane, 1, center_dimensions, scale)
                if out is None:
                    out = torch.empty(tuple(z.shape[:3]) + (n_col,), dtype=torch.float32, device=src.device)
                # ... and its pyramid interleaved into channel c of the result by the same kernel
                _runtime.cast_interleave(z.contiguous(), out, 1, 0, 1, n_col, c, z.numel())
            return out
        planes = [image_to_zoom_tensor(np.ascontiguousarray(frames[0][..., c:c + 1]), 1, center_dimensions, scale)
                  for c in range(n_col)]
        return np.concatenate(planes, axis=-1)
    levels = reference_levels(tuple(frames.shape[1:3]), center_dimensions, scale)
    packed = _plan(tuple(frames.shape[1:]), levels, dev).run(frames)
    h, w = packed.extents[0]
    return packed.data.reshape(len(levels), h, w, int(num_colors))


def classic_pyramid(image, scale, n_levels):
    """[H, W, C] or [n, H, W, C] -> PackedPyramid of whole-frame levels (BASELINE configs 1-5)."""
    frames, dev = _frames(image)
    levels = classic_levels(tuple(frames.shape[1:3]), scale, n_levels)
    return _plan(tuple(frames.shape[1:]), levels, dev).run(frames)
