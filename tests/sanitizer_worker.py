"""Worker of tests/test_sanitizers.py: runs INSIDE a python started with LD_PRELOAD=<asan runtime> and
SILENT_LIB_PATH=pysilent_amd/lib/libsilent_hostonly_asan.so (the host side of the library, silent_unity.hip, kernel launches compiled out,
device memory = host memory; pysilent_amd/csrc/silent_host_shim.h).  No GPU, no oracle: what is checked is that every line of
host code -- validation, tile / region / tap tables, row programs, walk plans, weight-stream packing, workspace layout, staging
of the host-pointer twins -- runs clean under AddressSanitizer + UndefinedBehaviorSanitizer for fuzzed geometries, that bad
arguments come back as error codes, and that an exception anywhere inside an entry point comes back as SILENT_E_NOMEM /
SILENT_E_INVALID instead of crossing the C ABI.

    python tests/sanitizer_worker.py [seed=0] [seconds=25]
"""
import ctypes as C
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from pysilent_amd import _lib, _runtime as rt  # noqa: E402
from pysilent_amd import constant_convolutions as cc  # noqa: E402
from pysilent_amd.pipeline import default_constants  # noqa: E402
from pysilent_amd.util.zoom.from_image import classic_levels, reference_levels  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 25.0
rng = np.random.default_rng(seed)
lib = _lib.load()
assert "hostonly" in _lib.LIB_PATH, "this worker must never run against the product library"
lib.silent_host_arm_fault.argtypes = [C.c_long]
lib.silent_host_arm_fault.restype = None
lib.silent_host_fail_new_after.argtypes = [C.c_long]
lib.silent_host_fail_new_after.restype = None
ctx = rt.get_context(0)
assert "host-only" in ctx.name
RGB = default_constants("rgb")
GRAY = default_constants("gray", 4)
stats = {"plans": 0, "ops": 0, "rejected": 0, "faults": 0, "new_faults": 0}


def extents(n_levels, lo=3, hi=70):
    return [(int(rng.integers(lo, hi)), int(rng.integers(lo, hi))) for _ in range(n_levels)]


def packed(ext, c, n_frames):
    px = sum(h * w for h, w in ext)
    return rt.PackedPyramid(rng.random(n_frames * px * c, dtype=np.float32) * 255.0, ext, c, n_frames)


def one_plan():
    h, w, c = int(rng.integers(8, 160)), int(rng.integers(8, 200)), int(rng.choice([1, 3]))
    if rng.random() < 0.5:
        # (ratios below e ** .5: the stream kernels' dense slot layout, the walk's 28 / 24-pixel tiles; up to 10 levels: ladders the
        # single-read paths refuse)
        levels = classic_levels((h, w), float(rng.choice([2.0, 1.6, math.e ** .5, 2 ** .5, 1.3, 1.2, 3.0])), int(rng.integers(1, 11)))
    else:
        cw, ch = int(rng.integers(4, max(5, w // 2 + 1))), int(rng.integers(4, max(5, h // 2 + 1)))
        levels = reference_levels((h, w), (cw, ch), float(rng.choice([2.0, math.e ** .5, 1.5])))
        if not levels:
            return
    if c == 3 and w % 4 and rng.random() < 0.5:
        w -= w % 4
        if w < 8:
            return
        levels = classic_levels((h, w), 2.0, int(rng.integers(1, 6)))
    plan = rt.PyramidPlan(h, w, c, levels, 0)
    stats["plans"] += 1
    _ = plan.streamable, plan.walk_plans
    frames = rng.random((int(rng.integers(1, 3)), h, w, c), dtype=np.float32)
    plan.run(frames)
    if c == 1:
        K = int(rng.choice([3, 4, 8]))
        plan.gray_pass(frames, GRAY["cs"], cc.end_bank(K).astype(np.float32))
    plan.close()


def one_op():
    n_levels, n_frames = int(rng.integers(1, 7)), int(rng.integers(1, 3))
    ext = extents(n_levels)
    which = int(rng.integers(0, 16))
    stats["ops"] += 1
    if which == 0:
        kh = int(rng.choice([1, 3, 5, 7]))
        ci, co = int(rng.choice([1, 3])), int(rng.choice([1, 3, 4, 8]))
        rt.conv2d_same(packed(ext, ci, n_frames), rng.standard_normal((kh, kh, ci, co)), relu=bool(rng.integers(2)),
                       clip_hi=255.0 if rng.integers(2) else None)
    elif which == 1:
        c = int(rng.choice([1, 3]))
        k = int(rng.choice([3, 7]))
        blur = rng.random((k, k, c, c)) if rng.integers(2) else np.broadcast_to(rng.random((k, k, 1, 1)), (k, k, c, c)).copy()
        rt.regulate(packed(ext, c, n_frames), blur, 1.0, 0.1, str(rng.choice(["ieee", "zero"])))
    elif which == 2:
        K = int(rng.choice([3, 4, 8]))
        rt.gray_line_end(packed(ext, 1, n_frames), GRAY["cs"], cc.end_bank(K))
    elif which == 3:
        p = [int(v) for v in rng.integers(0, 4, 4)]
        rt.pad_inwards(packed(ext, int(rng.choice([1, 3])), n_frames), *p)
    elif which == 4:
        x = packed(ext, 3, n_frames)
        rt.value_from_color(x), rt.bw_from_color(x), rt.nms3x3(x, str(rng.choice(["product", "fired"])))
    elif which == 5:
        x = packed(ext, 3, n_frames)
        rt.top_value_points(x, float(rng.random()), rt.value_from_color(x) if rng.integers(2) else None)
    elif which == 6:
        # regions from one window per axis to more than four (the separable prefix / suffix path)
        regions = [(max(1, h // int(rng.integers(1, 9))), max(1, w // int(rng.integers(1, 9)))) for h, w in ext]
        rt.max_value_indices_region(packed(ext, 1, n_frames), regions, None if rng.integers(2) else int(rng.integers(1, 50)))
    elif which == 7:
        x = packed(ext, int(rng.choice([1, 3])), n_frames)
        rt.select_peaks(x, 0.1)
    elif which == 8:
        rt.centroids(packed(ext, 1, n_frames), int(rng.integers(1, 6)), int(rng.integers(1, 6)))
    elif which == 9:
        x = packed(ext, 1, n_frames)
        rt.boosting_step(x, packed(ext, 1, n_frames), visualize=bool(rng.integers(2)))
    elif which == 10:
        rt.affine_clip(packed(ext, int(rng.choice([1, 3])), n_frames), 2.0, 1.0, 0.0, 255.0)
    elif which == 11:
        rt.resize_nearest(packed(ext, int(rng.choice([1, 3])), n_frames), extents(n_levels))
    elif which == 12:
        ks = dict(RGB)
        if rng.random() < 0.3:     # off the reference's structure: the basic / dense instantiations, a non-uniform blur
            name = str(rng.choice(["rgc", "rgby", "stripe", "blur", "end"]))
            ks[name] = ks[name] + rng.standard_normal(ks[name].shape).astype(np.float32) * 0.1
        rt.rgb_line_end(packed(ext, 3, n_frames), ks, flat_policy=str(rng.choice(["ieee", "zero"])))
    elif which == 13:
        rgb_keypoints_host(ext, n_frames)
    elif which == 14:
        n_px, cin, cout = int(rng.integers(1, 500)), int(rng.integers(1, 5)), int(rng.integers(1, 5))
        count = int(rng.integers(1, min(cin, cout) + 1))
        io, oo = int(rng.integers(0, cin - count + 1)), int(rng.integers(0, cout - count + 1))
        dt, code = [(np.uint8, _lib.DT_U8), (np.float64, _lib.DT_F64), (np.int16, _lib.DT_I16), (np.int64, _lib.DT_I64)][int(rng.integers(4))]
        # buffers that END with the last used element (not a whole stride)
        src = np.zeros((n_px - 1) * cin + io + count, dt)
        dst = np.zeros((n_px - 1) * cout + oo + count, np.float32)
        ctx.check(lib.silent_cast_interleave(ctx.handle, src.ctypes.data, code, n_px, cin, io, count, dst.ctypes.data, cout, oo))
    else:
        fp = C.POINTER(C.c_float)
        ks = {k: np.ascontiguousarray(v, np.float32) for k, v in RGB.items()}
        if rng.random() < 0.5:
            ks["end"] = ks["end"] + rng.standard_normal(ks["end"].shape).astype(np.float32)
        prm = _lib.RgbChainParams(*[ks[k].ctypes.data_as(fp) for k in ("rgc", "rgby", "stripe", "blur", "end")], 1.0, 0.1, 0, 255.0, 2)
        flags, masks = C.c_uint(0), (C.c_uint * 6)()
        assert lib.silent_rgb_chain_structure(C.byref(prm), C.byref(flags), masks) == 0
        buf = np.zeros(1024, np.float32)
        n, blocks = C.c_int(0), C.c_int(0)
        lib.silent_rgb_chain_stream(C.byref(prm), flags.value, buf.ctypes.data, C.byref(n), C.byref(blocks))


def rgb_keypoints_host(ext, n_frames, cap=None):
    fp = C.POINTER(C.c_float)
    ks = {k: np.ascontiguousarray(v, np.float32) for k, v in RGB.items()}
    prm = _lib.RgbChainParams(*[ks[k].ctypes.data_as(fp) for k in ("rgc", "rgby", "stripe", "blur", "end")], 1.0, 0.1, 0, 255.0, 2)
    x = packed(ext, 3, n_frames)
    px = x.frame_px * n_frames
    levels = (_lib.Extent * len(ext))(*[_lib.Extent(h, w) for h, w in ext])
    regions = (_lib.Extent * len(ext))(*[_lib.Extent(max(h // 2, 1), max(w // 2, 1)) for h, w in ext])
    orient, line = np.zeros(px * 3, np.float32), np.zeros(px * 3, np.float32)
    cap = x.frame_px if cap is None else cap
    idx, counts = np.zeros((n_frames, max(cap, 1), 4), np.int64), np.zeros(n_frames, np.int64)
    pv = np.zeros(px, np.float32) if rng.integers(2) else None
    rc = lib.silent_rgb_keypoints(ctx.handle, x.data.ctypes.data, levels, len(ext), n_frames, C.byref(prm), 0.1, regions,
                                  orient.ctypes.data, line.ctypes.data, None, pv.ctypes.data if pv is not None else None,
                                  idx.ctypes.data, cap, counts.ctypes.data)
    assert rc in (_lib.SILENT_OK, _lib.SILENT_E_CAPACITY), _lib.last_error(ctx.handle)


def displayer_frames(n_frames=3, native_dtype=None):
    """silent_displayer_*: one camera frame per call -- create, a few steps (eager, two captures, a replay), state, destroy."""
    h, w = int(rng.integers(40, 120)), int(rng.integers(40, 160))
    dt = native_dtype or [np.uint8, np.float32, np.uint16][int(rng.integers(3))]
    out = (int(rng.integers(8, w // 2)), int(rng.integers(8, h // 2)))
    d = rt.FrameDisplayer((h, w, 3), dt, out, float(rng.choice([math.e ** .5, 2.0, 1.6])), RGB, device=0)
    for _ in range(n_frames):
        res = d.step((rng.random((h, w, 3)) * 255).astype(dt))
        assert len(res) == 6 and res[0].shape == d.shapes[0]
    st = d.get_state()
    d.set_state(st)
    d.close()


def bad_arguments():
    """Every one of these must come back as an error code (ValueError / TypeError in Python), never as a crash."""
    ext = [(9, 11)]
    x = packed(ext, 3, 1)
    lv = (_lib.Extent * 1)(_lib.Extent(9, 11))
    out = np.zeros(9 * 11 * 3, np.float32)
    calls = [
        lambda: lib.silent_value_from_color(ctx.handle, x.data.ctypes.data, None, 1, 1, 3, out.ctypes.data),
        lambda: lib.silent_value_from_color(ctx.handle, x.data.ctypes.data, lv, 0, 1, 3, out.ctypes.data),
        lambda: lib.silent_value_from_color(ctx.handle, x.data.ctypes.data, lv, 17, 1, 3, out.ctypes.data),
        lambda: lib.silent_value_from_color(ctx.handle, x.data.ctypes.data, lv, 1, 0, 3, out.ctypes.data),
        lambda: lib.silent_value_from_color(ctx.handle, None, lv, 1, 1, 3, out.ctypes.data),
        lambda: lib.silent_value_from_color(None, x.data.ctypes.data, lv, 1, 1, 3, out.ctypes.data),
        lambda: lib.silent_nms3x3(ctx.handle, x.data.ctypes.data, (_lib.Extent * 1)(_lib.Extent(-4, 11)), 1, 1, 3, 0, out.ctypes.data),
        lambda: lib.silent_rgb_keypoints(ctx.handle, x.data.ctypes.data, None, 1, 1, None, 0.1, None, None, None, None, None, None, 0, None),
        lambda: lib.silent_pyramid_plan_create(ctx.handle, 0, 10, 1, None, 1, None),
        lambda: lib.silent_set_tuning(ctx.handle, 99, 0),
        lambda: lib.silent_cast_interleave(ctx.handle, x.data.ctypes.data, 42, 4, 1, 0, 1, out.ctypes.data, 1, 0),
        lambda: lib.silent_cast_interleave(ctx.handle, x.data.ctypes.data, 1, 4, 1, 0, 2, out.ctypes.data, 1, 0),
    ]
    for f in calls:
        rc = f()
        assert rc < 0, "a bad argument was accepted (rc %d)" % rc
        stats["rejected"] += 1
    # level-count / NULL checks that used to dereference first: silent_rgb_keypoints_dev with bad levels
    fp = C.POINTER(C.c_float)
    ks = {k: np.ascontiguousarray(v, np.float32) for k, v in RGB.items()}
    prm = _lib.RgbChainParams(*[ks[k].ctypes.data_as(fp) for k in ("rgc", "rgby", "stripe", "blur", "end")], 1.0, 0.1, 0, 255.0, 2)
    cnt = np.zeros(1, np.int64)
    for levels, n in ((None, 1), (lv, 0), (lv, 10 ** 6)):
        rc = lib.silent_rgb_keypoints_dev(ctx.handle, x.data.ctypes.data, levels, n, 1, C.byref(prm), 0.1, lv, None, out.ctypes.data,
                                          None, None, None, 0, cnt.ctypes.data, None)
        assert rc == _lib.SILENT_E_INVALID, rc
        stats["rejected"] += 1


def exception_barrier():
    """An exception inside ANY entry point must come back as a status code.  (a) silent_host_arm_fault(n): the n-th NEED_CTX from
    now on throws std::bad_alloc -- n = 1 is the entry point itself, n = 2 the *_dev twin a host-pointer form calls; (b)
    silent_host_fail_new_after(n): the n-th operator new of the library throws, through plan creation and the planners."""
    ext = [(20, 24), (10, 12)]
    x3, x1 = packed(ext, 3, 1), packed(ext, 1, 1)
    regions = [(10, 12), (5, 6)]
    calls = {
        "conv2d_same": lambda: rt.conv2d_same(x3, RGB["rgc"], relu=True),
        "regulate": lambda: rt.regulate(x3, RGB["blur"], 1.0, 0.1),
        "gray_line_end": lambda: rt.gray_line_end(x1, GRAY["cs"], GRAY["end"]),
        "pad_inwards": lambda: rt.pad_inwards(x3, 2, 2, 2, 2),
        "value_from_color": lambda: rt.value_from_color(x3),
        "bw_from_color": lambda: rt.bw_from_color(x3),
        "nms3x3": lambda: rt.nms3x3(x3),
        "top_value_points": lambda: rt.top_value_points(x3),
        "max_value_indices_region": lambda: rt.max_value_indices_region(x1, regions),
        "select_peaks": lambda: rt.select_peaks(x3),
        "centroids": lambda: rt.centroids(x1, 3, 3),
        "boosting_step": lambda: rt.boosting_step(x1, packed(ext, 1, 1)),
        "affine_clip": lambda: rt.affine_clip(x3),
        "resize_nearest": lambda: rt.resize_nearest(x3, [(7, 9), (3, 5)]),
        "rgb_line_end": lambda: rt.rgb_line_end(x3, RGB),
        "rgb_keypoints": lambda: rgb_keypoints_checked(ext),
        "pyramid_plan_create": lambda: rt.PyramidPlan(40, 48, 3, classic_levels((40, 48), 2.0, 3), 0).close(),
        "pyramid": lambda: run_plan(1),
        "gray_pass": lambda: run_plan(2),
        "set_profiling": lambda: ctx.check(lib.silent_set_profiling(ctx.handle, 1)),
        "malloc": lambda: ctx.check(lib.silent_malloc(ctx.handle, 64, C.byref(C.c_void_p()))),
        "synchronize": lambda: ctx.check(lib.silent_synchronize(ctx.handle, None)),
        "sparse_tail_stats": lambda: ctx.check(lib.silent_sparse_tail_stats(ctx.handle, (C.c_int64 * 5)())),
        "busy_wait": lambda: ctx.check(lib.silent_busy_wait_dev(ctx.handle, 5, None)),
        "displayer": lambda: displayer_frames(4, np.uint8),
    }
    for name, call in calls.items():
        call()                                   # works
        for n in (1, 2):
            lib.silent_host_arm_fault(n)
            try:
                call()
                hit = False
            except MemoryError as e:
                hit = True
                assert "out of host memory" in str(e), (name, str(e))
            finally:
                lib.silent_host_arm_fault(0)
            assert hit or n == 2, "%s: the armed fault did not surface as SILENT_E_NOMEM" % name
            stats["faults"] += int(hit)
        call()                                   # and works again afterwards
    # (b) real allocation failures, one after the other, through plan creation (tap tables, row programs, walk plans) and
    # through a keypoint call (region tables, std::string of error paths): NOMEM or success, nothing else, and no crash
    for what in ("plan3", "plan1", "keypoints", "displayer"):
        for k in range(1, 400):
            lib.silent_host_fail_new_after(k)
            try:
                if what == "plan3":
                    rt.PyramidPlan(64, 96, 3, classic_levels((64, 96), 2.0, 4), 0).close()
                elif what == "plan1":
                    rt.PyramidPlan(64, 96, 1, reference_levels((64, 96), (24, 16), math.e ** .5), 0).close()
                elif what == "displayer":
                    displayer_frames(2, np.uint8)
                else:
                    rt.max_value_indices_region(x1, [(2, 2), (1, 1)])
                done = True
            except MemoryError:
                done = False
                stats["new_faults"] += 1
            finally:
                lib.silent_host_fail_new_after(0)
            if done:
                break
        assert done, "%s never succeeded within 400 allocations" % what


def rgb_keypoints_checked(ext):
    fp = C.POINTER(C.c_float)
    ks = {k: np.ascontiguousarray(v, np.float32) for k, v in RGB.items()}
    prm = _lib.RgbChainParams(*[ks[k].ctypes.data_as(fp) for k in ("rgc", "rgby", "stripe", "blur", "end")], 1.0, 0.1, 0, 255.0, 2)
    x = packed(ext, 3, 1)
    levels = (_lib.Extent * len(ext))(*[_lib.Extent(h, w) for h, w in ext])
    regions = (_lib.Extent * len(ext))(*[_lib.Extent(max(h // 2, 1), max(w // 2, 1)) for h, w in ext])
    line = np.zeros(x.frame_px * 3, np.float32)
    idx, counts = np.zeros((1, x.frame_px, 4), np.int64), np.zeros(1, np.int64)
    rc = lib.silent_rgb_keypoints(ctx.handle, x.data.ctypes.data, levels, len(ext), 1, C.byref(prm), 0.1, regions, None,
                                  line.ctypes.data, None, None, idx.ctypes.data, x.frame_px, counts.ctypes.data)
    if rc != _lib.SILENT_E_CAPACITY:
        ctx.check(rc)


def run_plan(c):
    if c == 1:
        plan = rt.PyramidPlan(40, 48, 3, classic_levels((40, 48), 2.0, 3), 0)
        plan.run(np.zeros((1, 40, 48, 3), np.float32))
    else:
        plan = rt.PyramidPlan(40, 48, 1, classic_levels((40, 48), 2.0, 3), 0)
        plan.gray_pass(np.zeros((1, 40, 48, 1), np.float32), GRAY["cs"], GRAY["end"])
    plan.close()


def plan_eligibility():
    """Which plans get the single-read kernels is decided by host code alone (silent_pyramid_plan_create): the table of
    INTEGRATION.md section 4, checked here without a GPU (and under the sanitizers)."""
    def plan(shape, scale, n, c):
        p = rt.PyramidPlan(shape[0], shape[1], c, classic_levels(shape, scale, n), 0)
        r = (p.streamable, p.walk_plans)
        p.close()
        return r
    assert plan((270, 480), 2.0, 6, 3) == (False, (1, 36))
    assert plan((270, 480), math.e ** .5, 5, 3) == (False, (1, 32))
    assert plan((270, 480), 2 ** .5, 6, 3) == (False, (1, 28))                 # round 5: zoom steps below 1.6
    assert plan((200, 300), 2 ** (1 / 3), 5, 3) == (False, (1, 24))
    assert plan((270, 482), 2.0, 4, 3) == (False, (1, 36))                      # round 5: widths that are not a multiple of 4
    assert plan((270, 480), 2 ** .5, 10, 3) == (False, (0, 0))                  # more than 7 general levels: unit + region kernels
    assert plan((270, 480), 2.0, 5, 1) == (True, (0, 0))
    assert plan((270, 480), 2 ** .5, 8, 1) == (True, (0, 0))                    # round 5: the dense slot layout of the stream kernels
    assert plan((100, 260), 1.15, 3, 1) == (False, (0, 0))
    ref = rt.PyramidPlan(1080, 1920, 3, reference_levels((1080, 1920), (288, 192), math.e ** .5), 0)
    assert ref.walk_plans == (2, 32)                                            # a plan for the unit level + one union plan
    ref.close()
    stats["plans"] += 10


t0 = time.time()
bad_arguments()
exception_barrier()
plan_eligibility()
while time.time() - t0 < budget:
    try:
        r = rng.random()
        if r < 0.04:
            displayer_frames()
        elif r < 0.37:
            one_plan()
        else:
            one_op()
    except ValueError:
        stats["rejected"] += 1          # an unsupported geometry said so
print("sanitizer worker ok: %s in %.1f s" % (stats, time.time() - t0))
