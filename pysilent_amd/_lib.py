"""ctypes binding of libsilent_hip.so (include/silent_hip.h) -- the only way the package reaches the GPU.

There is no CPU fallback anywhere in this package: if the shared library is missing, or no
gfx950 device is visible, the filters raise.  (The CPU oracle lives in ``oracle/`` and is test
infrastructure; nothing under ``pysilent_amd`` imports it.)
"""
import ctypes as C
import importlib.util
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SILENT_LIB_PATH: load another build of the same library (kernel experiments: build.py --out ... -D..., scripts/ab_same_buffers.py)
LIB_PATH = os.environ.get("SILENT_LIB_PATH") or os.path.join(_HERE, "lib", "libsilent_hip.so")

SILENT_OK = 0
SILENT_E_INVALID = -1
SILENT_E_HIP = -2
SILENT_E_CAPACITY = -3
SILENT_E_UNSUPPORTED = -4
SILENT_E_NOMEM = -5

RELU = 1
CLIP = 2
FLAT_IEEE = 0
FLAT_ZERO = 1
NMS_PRODUCT = 0
NMS_FIRED = 1
RECOVERY_CONSTANT = 1
RECOVERY_INPUT = 2
MAX_LEVELS = 16
TUNE_GRAY, TUNE_RGB, TUNE_PYRAMID = 0, 1, 2
GRAY_PART_PYRAMID, GRAY_PART_FILTER = 1, 2
DT_U8, DT_F32, DT_F64, DT_I32, DT_U16, DT_I16, DT_I64 = 0, 1, 2, 3, 4, 5, 6
ABI_VERSION = 4


class Extent(C.Structure):
    _fields_ = [("h", C.c_int32), ("w", C.c_int32)]


class PyrLevel(C.Structure):
    _fields_ = [("src_y0", C.c_int32), ("src_x0", C.c_int32), ("src_h", C.c_int32), ("src_w", C.c_int32),
                ("zoom_h", C.c_int32), ("zoom_w", C.c_int32), ("out_h", C.c_int32), ("out_w", C.c_int32)]


class RgbChainParams(C.Structure):
    _fields_ = [("rgc", C.POINTER(C.c_float)), ("rgby", C.POINTER(C.c_float)), ("stripe", C.POINTER(C.c_float)),
                ("blur", C.POINTER(C.c_float)), ("end", C.POINTER(C.c_float)),
                ("regulation_value", C.c_float), ("regulation_root", C.c_float), ("flat_policy", C.c_int32),
                ("clip_hi", C.c_float), ("pad", C.c_int32)]


class BoostingParams(C.Structure):
    _fields_ = [("exhaustion_max", C.c_float), ("excitation_max", C.c_float), ("recovery_mode", C.c_uint),
                ("recovery_amount", C.c_float), ("recovery_percentage", C.c_float), ("visualize", C.c_int32)]


class AffineParams(C.Structure):
    _fields_ = [("mul", C.c_float), ("div", C.c_float), ("add", C.c_float), ("lo", C.c_float), ("hi", C.c_float),
                ("post_add", C.c_float)]


class DisplayerParams(C.Structure):
    """silent_displayer_params (include/silent_hip.h)."""
    _fields_ = [("frame_h", C.c_int32), ("frame_w", C.c_int32), ("frame_dtype", C.c_int32), ("centroid_region_h", C.c_int32),
                ("centroid_region_w", C.c_int32), ("chain", RgbChainParams), ("boosting", BoostingParams)]


class SilentLibraryError(RuntimeError):
    """libsilent_hip.so is missing or cannot be loaded."""


_lib = None

_vp = C.c_void_p
_fp = C.c_void_p          # float* passed as an address (host ndarray.ctypes.data or a device pointer)
_ep = C.POINTER(Extent)
_i, _u, _f, _d, _sz = C.c_int, C.c_uint, C.c_float, C.c_double, C.c_size_t

# name -> argtypes (restype is int unless listed in _RESTYPES)
_SIGNATURES = {
    "silent_abi_version": [],
    "silent_device_count": [C.POINTER(_i)],
    "silent_create": [_i, C.POINTER(_vp)],
    "silent_destroy": [_vp],
    "silent_last_error": [_vp],
    "silent_device_name": [_vp, C.c_char_p, _sz],
    "silent_malloc": [_vp, _sz, C.POINTER(_vp)],
    "silent_free": [_vp, _vp],
    "silent_memcpy_h2d": [_vp, _vp, _vp, _sz, _vp],
    "silent_memcpy_d2h": [_vp, _vp, _vp, _sz, _vp],
    "silent_synchronize": [_vp, _vp],
    "silent_busy_wait_dev": [_vp, _u, _vp],
    "silent_trace_marker_dev": [_vp, _vp],
    "silent_pyramid_plan_create": [_vp, _i, _i, _i, C.POINTER(PyrLevel), _i, C.POINTER(_vp)],
    "silent_pyramid_plan_destroy": [_vp],
    "silent_pyramid": [_vp, _vp, _fp, _i, _fp],
    "silent_pyramid_dev": [_vp, _vp, _fp, _i, _fp, _vp],
    "silent_conv2d_same": [_vp, _fp, _ep, _i, _i, _i, _fp, _i, _i, _i, _u, _f, _fp],
    "silent_conv2d_same_dev": [_vp, _fp, _ep, _i, _i, _i, _fp, _i, _i, _i, _u, _f, _fp, _vp],
    "silent_gray_line_end": [_vp, _fp, _ep, _i, _i, _fp, _fp, _i, _f, _fp, _fp],
    "silent_gray_line_end_dev": [_vp, _fp, _ep, _i, _i, _fp, _fp, _i, _f, _fp, _fp, _vp],
    "silent_gray_pass": [_vp, _vp, _fp, _i, _fp, _fp, _i, _f, _fp, _fp, _fp],
    "silent_gray_pass_dev": [_vp, _vp, _fp, _i, _fp, _fp, _i, _f, _fp, _fp, _fp, _vp],
    "silent_gray_pass_parts_dev": [_vp, _vp, _fp, _i, _fp, _fp, _i, _f, _fp, _fp, _fp, _u, _vp],
    "silent_pyramid_plan_is_streamable": [_vp],
    "silent_pyramid_plan_walk_plans": [_vp, C.POINTER(C.c_int)],
    "silent_gather_d2h": [_vp, _vp, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), _i, _vp],
    "silent_set_profiling": [_vp, _i],
    "silent_set_tuning": [_vp, _i, _u],
    "silent_get_tuning": [_vp, _i, C.POINTER(C.c_uint)],
    "silent_profile_elapsed_ms": [_vp, C.POINTER(_f), C.POINTER(C.c_int64)],
    "silent_regulate": [_vp, _fp, _ep, _i, _i, _i, _fp, _i, _i, _f, _f, _i, _fp],
    "silent_regulate_dev": [_vp, _fp, _ep, _i, _i, _i, _fp, _i, _i, _f, _f, _i, _fp, _vp],
    "silent_pad_inwards": [_vp, _fp, _ep, _i, _i, _i, _i, _i, _i, _i, _fp],
    "silent_pad_inwards_dev": [_vp, _fp, _ep, _i, _i, _i, _i, _i, _i, _i, _fp, _vp],
    "silent_value_from_color": [_vp, _fp, _ep, _i, _i, _i, _fp],
    "silent_value_from_color_dev": [_vp, _fp, _ep, _i, _i, _i, _fp, _vp],
    "silent_bw_from_color": [_vp, _fp, _ep, _i, _i, _i, _fp],
    "silent_bw_from_color_dev": [_vp, _fp, _ep, _i, _i, _i, _fp, _vp],
    "silent_nms3x3": [_vp, _fp, _ep, _i, _i, _i, _i, _fp],
    "silent_nms3x3_dev": [_vp, _fp, _ep, _i, _i, _i, _i, _fp, _vp],
    "silent_top_value_points": [_vp, _fp, _fp, _ep, _i, _i, _i, _d, _fp],
    "silent_top_value_points_dev": [_vp, _fp, _fp, _ep, _i, _i, _i, _d, _fp, _vp],
    "silent_max_value_indices_region": [_vp, _fp, _ep, _i, _i, _ep, _vp, _sz, _vp],
    "silent_max_value_indices_region_dev": [_vp, _fp, _ep, _i, _i, _ep, _vp, _sz, _vp, _vp],
    "silent_select_peaks": [_vp, _fp, _fp, _ep, _i, _i, _i, _d, _fp, _fp, _fp],
    "silent_select_peaks_dev": [_vp, _fp, _fp, _ep, _i, _i, _i, _d, _fp, _fp, _fp, _vp],
    "silent_select_keypoints": [_vp, _fp, _fp, _ep, _i, _i, _i, _d, _ep, _fp, _vp, _sz, _vp],
    "silent_select_keypoints_dev": [_vp, _fp, _fp, _ep, _i, _i, _i, _d, _ep, _fp, _vp, _sz, _vp, _vp],
    "silent_centroids": [_vp, _fp, _ep, _i, _i, _i, _i, _fp, _fp],
    "silent_centroids_dev": [_vp, _fp, _ep, _i, _i, _i, _i, _fp, _fp, _vp],
    "silent_boosting_step": [_vp, _fp, _ep, _i, _i, C.POINTER(BoostingParams), _fp, _fp, _fp],
    "silent_boosting_step_dev": [_vp, _fp, _ep, _i, _i, C.POINTER(BoostingParams), _fp, _fp, _fp, _vp],
    "silent_affine_clip": [_vp, _fp, _sz, C.POINTER(AffineParams), _fp],
    "silent_affine_clip_dev": [_vp, _fp, _sz, C.POINTER(AffineParams), _fp, _vp],
    "silent_cast_interleave": [_vp, _vp, _i, _sz, _i, _i, _i, _fp, _i, _i],
    "silent_cast_interleave_dev": [_vp, _vp, _i, _sz, _i, _i, _i, _fp, _i, _i, _vp],
    "silent_resize_nearest": [_vp, _fp, _ep, _i, _i, _i, _ep, _fp],
    "silent_resize_nearest_dev": [_vp, _fp, _ep, _i, _i, _i, _ep, _fp, _vp],
    "silent_rgb_chain_structure": [C.POINTER(RgbChainParams), C.POINTER(C.c_uint), C.POINTER(C.c_uint)],
    "silent_rgb_keypoints": [_vp, _fp, _ep, _i, _i, C.POINTER(RgbChainParams), _d, _ep, _fp, _fp, _fp, _fp, _vp, _sz, _vp],
    "silent_rgb_keypoints_dev": [_vp, _fp, _ep, _i, _i, C.POINTER(RgbChainParams), _d, _ep, _fp, _fp, _fp, _fp, _vp, _sz, _vp, _vp],
    "silent_sparse_tail_stats": [_vp, C.POINTER(C.c_int64)],
    "silent_rgb_chain_stream": [C.POINTER(RgbChainParams), C.c_uint, _fp, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "silent_rgb_line_end": [_vp, _fp, _ep, _i, _i, C.POINTER(RgbChainParams), _fp, _fp, _fp],
    "silent_rgb_line_end_dev": [_vp, _fp, _ep, _i, _i, C.POINTER(RgbChainParams), _fp, _fp, _fp, _vp],
    "silent_displayer_create": [_vp, C.POINTER(DisplayerParams), C.POINTER(PyrLevel), _i, C.POINTER(_vp)],
    "silent_displayer_destroy": [_vp],
    "silent_displayer_shape": [_vp, C.POINTER(C.c_int32), C.POINTER(_sz)],
    "silent_displayer_step": [_vp, _vp, C.POINTER(_vp), C.POINTER(_f)],
    "silent_displayer_add_slot": [_vp, C.POINTER(_i)],
    "silent_displayer_step_slot": [_vp, _vp, _i, C.POINTER(_vp), C.POINTER(_f)],
    "silent_displayer_input": [_vp, C.POINTER(_vp), C.POINTER(_sz)],
    "silent_displayer_get_state": [_vp, _fp],
    "silent_displayer_set_state": [_vp, _fp],
}
_RESTYPES = {"silent_destroy": None, "silent_pyramid_plan_destroy": None, "silent_displayer_destroy": None, "silent_last_error": C.c_char_p}

EXPORTED_SYMBOLS = tuple(sorted(_SIGNATURES))


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so with the same
    SONAME as /opt/rocm's; if torch is installed, load ITS copy first (by path, RTLD_GLOBAL) so that our
    DT_NEEDED libamdhip64.so.7 binds to the very file ``import torch`` will map later.  Without torch the
    library's RUNPATH (/opt/rocm/lib) applies."""
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load (once) and return the ctypes handle; raises SilentLibraryError when the .so is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SilentLibraryError(
            "%s not found: build it with `python pysilent_amd/csrc/build.py` (hipcc --offload-arch=gfx950). "
            "pysilent_amd has no CPU fallback." % LIB_PATH)
    _preload_hip_runtime()
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise SilentLibraryError("cannot load %s: %s" % (LIB_PATH, e))
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header / library mismatch
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    got = lib.silent_abi_version()
    if got != ABI_VERSION:
        raise SilentLibraryError("libsilent_hip.so ABI %d != binding ABI %d" % (got, ABI_VERSION))
    _lib = lib
    return lib


def last_error(ctx_handle):
    msg = load().silent_last_error(ctx_handle)
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc, ctx_handle=None):
    """Map a silent_status to the Python exception the reference's callers would see."""
    if rc == SILENT_OK:
        return
    msg = last_error(ctx_handle) or ("silent_status %d" % rc)
    if rc in (SILENT_E_INVALID, SILENT_E_UNSUPPORTED, SILENT_E_CAPACITY):
        raise ValueError(msg)
    if rc == SILENT_E_NOMEM:
        raise MemoryError(msg)
    raise RuntimeError(msg)
