#!/usr/bin/env python3
"""Golden vectors for SURVEY.md section 8a-1 from the reference's OWN wrapper, `image_to_zoom_tensor`
(slam_recognition/util/zoom/from_image.py:10-69), executed unmodified.

Runs ONLY in the build container (needs /root/reference).  The function was written for NumPy < 1.23, where indexing with
a LIST of slices meant what a tuple means; on the NumPy of this image that raises IndexError.  Its text is
not touched: the module's two global names `np` and `ndimage` are replaced by pass-through proxies that hand out ndarray
VIEWS of a subclass whose __getitem__ / __setitem__ read a list of slices / None as a tuple -- the indexing rule the code
was written against -- and whose `empty` is zero-filled, so that canvas pixels the function never writes are defined
(they are uninitialised in the reference; the oracle and the product define them as 0).  scipy.ndimage.zoom itself is
the real one.  Inert `tensorflow` modules as in make_golden.py (the package imports it at module scope).

Stores float32 inputs and float64 outputs for a few small cases in tests/golden/pyramid.npz (two of them with NaN / inf pixels:
the reference's own outputs pin the footprint of a non-finite pixel, scipy's six taps per axis at zoom 1).
Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_pyramid.py
"""
import importlib
import math
import os
import sys
import types

import numpy as np
from scipy import ndimage as real_ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, _install_inert_tensorflow  # noqa: E402


def _key(k):
    if isinstance(k, list) and any(isinstance(e, slice) or e is None for e in k):
        return tuple(k)
    return k


class OldIndexing(np.ndarray):
    """ndarray with NumPy < 1.23's reading of a list of slices as an index tuple."""

    def __getitem__(self, k):
        return super().__getitem__(_key(k))

    def __setitem__(self, k, v):
        super().__setitem__(_key(k), v)


class _Proxy(types.ModuleType):
    def __init__(self, real, overrides):
        super().__init__(real.__name__)
        self._real, self._over = real, overrides

    def __getattr__(self, name):
        if name in self._over:
            return self._over[name]
        return getattr(self._real, name)


def main():
    sys.dont_write_bytecode = True
    _install_inert_tensorflow()
    sys.path.insert(0, REF)
    mod = importlib.import_module("slam_recognition.util.zoom.from_image")
    mod.np = _Proxy(np, {
        "empty": lambda shape, *a, **k: np.zeros(shape, *a, **k).view(OldIndexing),
        "squeeze": lambda a, *x, **k: np.squeeze(np.asarray(a), *x, **k),
    })
    mod.ndimage = _Proxy(real_ndimage, {
        "zoom": lambda *a, **k: real_ndimage.zoom(*a, **k).view(OldIndexing),
    })
    rng = np.random.default_rng(11)
    cases = {
        # name: (H, W, C, center_dimensions (w, h), scale)
        "default_small": (60, 80, 3, (36, 24), math.e ** 0.5),     # the reference default's proportions (640x480 -> 288x192)
        "two_to_one": (64, 96, 3, (24, 16), 2.0),
        "gray_three": (50, 70, 1, (20, 15), 1.5),
        "tall": (90, 40, 3, (16, 30), math.e ** 0.5),               # the crop clips on one axis first
        "odd": (37, 53, 2, (17, 11), 1.7),
    }
    out = {}
    for name, (h, w, c, center, scale) in cases.items():
        img = rng.integers(0, 256, (h, w, c)).astype(np.float32)
        z = mod.image_to_zoom_tensor(img.view(OldIndexing), c, list(center), scale)
        out[name + "_in"] = img
        out[name + "_out"] = np.asarray(z, np.float64)
        out[name + "_par"] = np.array([center[0], center[1], scale], np.float64)
        print("%-14s in %s -> %s" % (name, img.shape, z.shape))
    # Non-finite pixels (round 5): NaN / -NaN / +-inf at the corners, on the edges and inside the innermost crop (level 0 is that crop
    # at zoom 1: scipy's SIX taps carry each of them to a 6 x 6 block of outputs, rows / columns p - 3 .. p + 2), next to each other,
    # and outside it (only the outer levels see those).  Own generator: the cases above keep their values.
    rng2 = np.random.default_rng(12)
    neg_nan = np.frombuffer(np.uint32(0xffc00000).tobytes(), np.float32)[0]
    for name, (h, w, c, center, scale) in {"nonfinite_rgb": (60, 80, 3, (36, 24), math.e ** 0.5),
                                            "nonfinite_gray": (50, 70, 1, (20, 15), 1.5)}.items():
        img = rng2.integers(0, 256, (h, w, c)).astype(np.float32)
        y0, x0 = int(max((h - center[1]) / 2, 0)), int(max((w - center[0]) / 2, 0))   # the innermost crop (from_image.py:50-51)
        y1, x1 = int((h + center[1]) / 2) - 1, int((w + center[0]) / 2) - 1
        spots = [(y0, x0, np.nan), (y0, x1, np.inf), (y1, x0, -np.inf), (y1, x1, neg_nan),            # its four corners
                 (y0, (x0 + x1) // 2, np.inf), ((y0 + y1) // 2, x1, np.nan),                          # edges
                 ((y0 + y1) // 2, (x0 + x1) // 2, -np.inf), ((y0 + y1) // 2, (x0 + x1) // 2 + 1, np.inf),   # neighbours, opposite signs
                 (y0 + 4, x0 + 5, np.nan), (y1 - 3, x1 - 3, np.inf),                                  # 3 from the far edges: the sixth tap's reach
                 (1, 2, np.nan), (h - 2, w - 3, np.inf)]                                              # outside the innermost crop
        for i, (y, x, v) in enumerate(spots):
            img[y, x, i % c] = v
        with np.errstate(invalid="ignore"):
            z = mod.image_to_zoom_tensor(img.view(OldIndexing), c, list(center), scale)
        out[name + "_in"] = img
        out[name + "_out"] = np.asarray(z, np.float64)
        out[name + "_par"] = np.array([center[0], center[1], scale], np.float64)
        print("%-14s in %s -> %s, %d non-finite outputs in level 0" % (name, img.shape, z.shape, int((~np.isfinite(z[0])).sum())))
    # zoom_tensor_to_image_list (util/zoom/to_image_list.py:7-15), same indexing idiom, same shim: the display glue of f-3
    tl = importlib.import_module("slam_recognition.util.zoom.to_image_list")
    tl.np = mod.np
    for name in ("default_small", "gray_three"):
        imgs = tl.zoom_tensor_to_image_list(np.clip(out[name + "_out"], 0, 255).view(OldIndexing))
        out[name + "_images"] = np.stack([np.asarray(i) for i in imgs], 0)
        print("%-14s image list: %d x %s %s" % (name, len(imgs), imgs[0].shape, imgs[0].dtype))
    np.savez_compressed(os.path.join(HERE, "pyramid.npz"), **out)
    print("wrote", os.path.join(HERE, "pyramid.npz"), os.path.getsize(os.path.join(HERE, "pyramid.npz")), "bytes")


if __name__ == "__main__":
    main()
