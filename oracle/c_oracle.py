"""ctypes binding of oracle/libsilent_oracle.so (the C restatement) -- TEST INFRASTRUCTURE ONLY.

Same import restrictions as silent_oracle.py: tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg only.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SILENT_ORACLE_SO: another build of the same source (oracle/Makefile: `native` for bench.py's cpu_baseline leg on the box
# that times it, `asan` for tests/test_sanitizers.py)
_SO = os.environ.get("SILENT_ORACLE_SO") or os.path.join(_HERE, "libsilent_oracle.so")
_lib = None

_f = C.POINTER(C.c_float)


def build(force=False):
    if os.environ.get("SILENT_ORACLE_SO"):
        return _SO
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "silent_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.so_num_threads.restype = C.c_int
        L.so_max_value_indices_region.restype = C.c_int64
        L.so_max_value_indices_region.argtypes = [_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                  C.POINTER(C.c_int64), C.c_int64]
        L.so_conv2d_same.argtypes = [_f, C.c_int, C.c_int, C.c_int, C.c_int, _f, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_float, _f]
        L.so_regulate.argtypes = [_f, C.c_int, C.c_int, C.c_int, C.c_int, _f, C.c_int, C.c_int, C.c_float,
                                  C.c_float, C.c_int, _f]
        L.so_pad_inwards.argtypes = [_f] + [C.c_int] * 8 + [_f]
        L.so_value_from_color.argtypes = [_f, C.c_size_t, C.c_int, _f]
        L.so_nms3x3.argtypes = [_f] + [C.c_int] * 5 + [_f]
        L.so_top_value_points.argtypes = [_f, _f] + [C.c_int] * 4 + [C.c_double, _f]
        L.so_zoom_level.argtypes = [_f] + [C.c_int] * 11 + [_f]
        L.so_gray_line_end_level.argtypes = [_f, C.c_int, C.c_int, _f, _f, C.c_int, C.c_float, _f, _f]
        L.so_gray_pass_frames.restype = C.c_double
        L.so_gray_pass_frames.argtypes = [_f, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, _f, _f, C.c_int,
                                          C.c_float]
        L.so_rgb_pass_frames.restype = C.c_int64
        L.so_rgb_pass_frames.argtypes = [_f, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, _f, _f, _f, _f, _f,
                                         C.c_float, C.c_float, C.c_int, C.c_float, C.c_int, C.c_double,
                                         C.POINTER(C.c_int64), C.c_int64, C.POINTER(C.c_int64)]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_f)


def _c32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def num_threads():
    return lib().so_num_threads()


def set_num_threads(n):
    lib().so_set_num_threads(C.c_int(n))


def conv2d_same(x, k, relu=False, clip_hi=None):
    x, k = _c32(x), _c32(k)
    n, h, w, ci = x.shape
    kh, kw, ki, co = k.shape
    assert ki == ci and co <= 16
    out = np.empty((n, h, w, co), np.float32)
    flags = (1 if relu else 0) | (2 if clip_hi is not None else 0)
    lib().so_conv2d_same(_p(x), n, h, w, ci, _p(k), kh, kw, co, flags, float(clip_hi or 0.0), _p(out))
    return out


def regulate(x, blur, rv, root=0.5, flat_policy="ieee"):
    x, blur = _c32(x), _c32(blur)
    n, h, w, c = x.shape
    out = np.empty_like(x)
    lib().so_regulate(_p(x), n, h, w, c, _p(blur), blur.shape[0], blur.shape[1], float(rv), float(root),
                      {"ieee": 0, "zero": 1}[flat_policy], _p(out))
    return out


def pad_inwards(x, paddings):
    x = _c32(x)
    n, h, w, c = x.shape
    out = np.empty_like(x)
    lib().so_pad_inwards(_p(x), n, h, w, c, int(paddings[1][0]), int(paddings[1][1]), int(paddings[2][0]),
                         int(paddings[2][1]), _p(out))
    return out


def value_from_color(x):
    x = _c32(x)
    out = np.empty(x.shape[:-1] + (1,), np.float32)
    lib().so_value_from_color(_p(x), x.size // x.shape[-1], x.shape[-1], _p(out))
    return out


def nms3x3(x, mode="product"):
    x = _c32(x)
    n, h, w, c = x.shape
    out = np.empty_like(x)
    lib().so_nms3x3(_p(x), n, h, w, c, {"product": 0, "fired": 1}[mode], _p(out))
    return out


def top_value_points(color, top_percent=0.1, value=None):
    color = _c32(color)
    value = value_from_color(color) if value is None else _c32(value)
    n, h, w, c = color.shape
    out = np.empty_like(color)
    lib().so_top_value_points(_p(color), _p(value), n, h, w, c, float(top_percent), _p(out))
    return out


def max_value_indices_region(color, region_shape, value=None):
    value = value_from_color(_c32(color)) if value is None else _c32(value)
    n, h, w, _ = value.shape
    cap = n * h * w
    idx = np.empty((cap, 4), np.int64)
    cnt = lib().so_max_value_indices_region(_p(value), n, h, w, int(region_shape[1]), int(region_shape[2]),
                                            idx.ctypes.data_as(C.POINTER(C.c_int64)), cap)
    return idx[:cnt].copy()


def zoom_level(frame, y0, x0, ch, cw, zh, zw, oh, ow):
    frame = _c32(frame)
    H, W, Cc = frame.shape
    out = np.empty((oh, ow, Cc), np.float32)
    lib().so_zoom_level(_p(frame), H, W, Cc, y0, x0, ch, cw, zh, zw, oh, ow, _p(out))
    return out


def classic_pyramid(frame, extents):
    frame = _c32(frame)
    H, W, _ = frame.shape
    return [zoom_level(frame, 0, 0, H, W, zh, zw, zh, zw)[None] for zh, zw in extents]


def gray_line_end_level(lev, cs_k, end_k, clip_hi=255.0):
    lev, cs_k, end_k = _c32(lev), _c32(cs_k), _c32(end_k)
    _, h, w, _ = lev.shape
    K = end_k.shape[-1]
    cs = np.empty((1, h, w, 1), np.float32)
    end = np.empty((1, h, w, K), np.float32)
    lib().so_gray_line_end_level(_p(lev), h, w, _p(cs_k), _p(end_k), K, float(clip_hi), _p(cs), _p(end))
    return cs, end


def gray_pass_frames(frames, extents, cs_k, end_k, clip_hi=255.0):
    """Whole gray pass on a batch [B, H, W] of frames, one frame per OpenMP thread (bench.py's cpu_baseline leg).
    Returns the checksum the C side keeps the work alive with."""
    frames, cs_k, end_k = _c32(frames), _c32(cs_k), _c32(end_k)
    B, H, W = frames.shape[:3]
    ext = np.ascontiguousarray(np.asarray(extents, dtype=np.int32).reshape(-1, 2))
    return lib().so_gray_pass_frames(_p(frames), B, H, W, ext.ctypes.data_as(C.POINTER(C.c_int)), ext.shape[0],
                                     _p(cs_k), _p(end_k), end_k.shape[-1], float(clip_hi))


def rgb_pass_frames(frames, extents, kernels, flat_policy="ieee", top_percent=0.1, rv=1.0, root=0.1, clip_hi=255.0, pad=2,
                    cap=0):
    """BASELINE config 3 on a batch [B, H, W, 3] of frames, one frame per OpenMP thread: classic pyramid -> reference chain ->
    top-percent -> NMS -> value -> per-region keypoint indices (region = half the level).  Returns (counts [B] int64, rows):
    rows is None when cap == 0, else a list of int64 [K_f, 4] arrays (level, y, x, 0)."""
    frames = _c32(frames)
    B, H, W = frames.shape[:3]
    ext = np.ascontiguousarray(np.asarray(extents, dtype=np.int32).reshape(-1, 2))
    ks = [_c32(kernels[n]) for n in ("rgc", "rgby", "stripe", "blur", "end")]
    counts = np.zeros(B, np.int64)
    idx = np.empty((B, cap, 4), np.int64) if cap else None
    lib().so_rgb_pass_frames(_p(frames), B, H, W, ext.ctypes.data_as(C.POINTER(C.c_int)), ext.shape[0], *[_p(k) for k in ks],
                             float(rv), float(root), {"ieee": 0, "zero": 1}[flat_policy], float(clip_hi), int(pad),
                             float(top_percent), idx.ctypes.data_as(C.POINTER(C.c_int64)) if cap else None, int(cap),
                             counts.ctypes.data_as(C.POINTER(C.c_int64)))
    rows = [idx[f, :min(int(counts[f]), cap)].copy() for f in range(B)] if cap else None
    return counts, rows
