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

Stores float32 inputs and float64 outputs for a few small cases in tests/golden/pyramid.npz.
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
