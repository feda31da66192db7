"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Tolerances (BASELINE.json north_star): response maps within 1e-5 relative (see conftest.assert_close
for the exact statement), keypoint indices and every other integer / mask result bit-exact.
"""
import math

import numpy as np
import pytest

import err_bound as eb

import silent_oracle as so
import c_oracle as co
from conftest import assert_close, assert_regulated_close, noise_frame, structured_frame
from pysilent_amd._lib import TUNE_GRAY, TUNE_PYRAMID, TUNE_RGB  # noqa: F401

pytestmark = pytest.mark.gpu

RTOL = 1e-5


@pytest.fixture(scope="module")
def rt():
    from pysilent_amd import _runtime
    _runtime.get_context()          # fails loudly if there is no gfx950 device
    return _runtime


def f32(k):
    return np.asarray(k, np.float64).astype(np.float32)


def assert_gray_level_close(got_pyr, got_cs, got_end, want_pyr, want_cs, want_end, cs_kernel, bank, tag):
    """pyramid level, CS map and line-end maps of the gray pass against the oracle's: range-relative 1e-5 and, element by
    element, the propagated rounding bound (tests/err_bound.py)."""
    e_pyr = eb.zoom(want_pyr)
    e_cs, e_end = eb.gray_chain(want_pyr, cs_kernel, bank, want_cs, e_pyr)
    assert_close(got_pyr, want_pyr, RTOL, scale=255.0, what="pyramid " + tag, bound=e_pyr)
    assert_close(got_cs, want_cs, RTOL, scale=255.0, what="cs " + tag, bound=e_cs)
    assert_close(got_end, want_end, RTOL, scale=255.0, what="end " + tag, bound=e_end)


# ----------------------------------------------------------------------------- convolution

@pytest.mark.parametrize("name", ["rgc", "rgby", "stripe", "end", "blur"])
@pytest.mark.parametrize("shape", [(2, 37, 53, 3), (1, 5, 3, 3), (3, 16, 64, 3), (1, 1, 1, 3), (1, 17, 130, 3)])
def test_conv2d_reference_kernels(rt, kernels, name, shape):
    x = np.random.default_rng(7).integers(0, 256, shape).astype(np.float32)
    got = rt.conv2d_same(x, kernels[name], relu=True)
    assert_close(got, so.conv2d_same(x, kernels[name], relu=True), RTOL, what=name, bound=eb.conv(x, kernels[name]))


@pytest.mark.parametrize("gen", ["rgb_2d_edge_tensors", "rgb_2d_edge_tensors_time_diff", "rgb_2d_end_tensors"])
def test_conv2d_7x7_thick_edge_banks(rt, gen):
    """the 7x7x3x3 edge_tensor banks through the 7x7 stencil (the blur's kernel shape)."""
    from pysilent_amd.constant_convolutions import edge_orientation_detector as eod
    from pysilent_amd.util.apply_filter import apply_filter
    k = getattr(eod, gen)()
    assert k.shape == (7, 7, 3, 3)
    x = structured_frame(5, 61, 83, 3)[None]
    raw = so.conv2d_same(x, k)
    assert_close(apply_filter(x, k), raw, RTOL, what=gen, bound=eb.conv(x, k))
    # clipped outputs: the rounding noise of the 147-tap sums follows the magnitude of the UNCLIPPED responses
    assert_close(rt.conv2d_same(x, k, relu=True, clip_hi=255.0), so.conv2d_same(x, k, relu=True, clip_hi=255.0), RTOL,
                 scale=float(np.abs(raw).max()), what=gen + " relu clip", bound=eb.conv(x, k))


@pytest.mark.parametrize("kshape", [(3, 3, 1, 1), (3, 3, 1, 3), (3, 3, 1, 4), (3, 3, 1, 8), (3, 3, 3, 1), (3, 3, 3, 4),
                                    (7, 7, 1, 1), (5, 5, 3, 2), (2, 2, 3, 3), (1, 3, 3, 3), (4, 6, 2, 5)])
def test_conv2d_shapes_including_generic_path(rt, kshape):
    rng = np.random.default_rng(11)
    x = (rng.standard_normal((2, 23, 71, kshape[2])) * 40).astype(np.float32)
    k = rng.standard_normal(kshape)
    raw = so.conv2d_same(x, k)
    assert_close(rt.conv2d_same(x, k), raw, RTOL, what=str(kshape), bound=eb.conv(x, k))
    # clipped at 30 while the Gaussian-weight sums reach hundreds: the rounding noise follows the unclipped magnitude
    assert_close(rt.conv2d_same(x, k, relu=True, clip_hi=30.0), so.conv2d_same(x, k, relu=True, clip_hi=30.0), RTOL,
                 scale=float(np.abs(raw).max()), what=str(kshape) + " relu clip", bound=eb.conv(x, k))


def test_conv2d_known_answers(rt, kernels):
    # impulse: output = kernel flipped around the impulse (cross-correlation), independent of the oracle
    k = f32(kernels["end"])
    x = np.zeros((1, 9, 9, 3), np.float32)
    x[0, 4, 4, 1] = 1.0
    out = rt.conv2d_same(x, k)
    for dy in range(3):
        for dx in range(3):
            np.testing.assert_array_equal(out[0, 4 - (dy - 1), 4 - (dx - 1)], k[dy, dx, 1])
    # constant image: interior = c * sum of taps (midget_rgc: 4/3 - 2/3 per diagonal channel); border from SAME zeros
    rgc = f32(kernels["rgc"])
    x = np.full((1, 8, 8, 3), 30.0, np.float32)
    out = rt.conv2d_same(x, rgc)
    np.testing.assert_allclose(out[0, 1:-1, 1:-1], 30.0 * 2.0 / 3.0, rtol=1e-6)
    corner = 30.0 * rgc[1:, 1:, 0, 0].astype(np.float64).sum()
    np.testing.assert_allclose(out[0, 0, 0, 0], corner, rtol=1e-6)


def test_filters_api_matches_reference_chain(rt, kernels):
    from pysilent_amd import filters
    from pysilent_amd.util.apply_filter import apply_filter
    x = noise_frame(3, 48, 64, 3)[None]
    rgc = filters.rgc_filter(x)
    assert_close(rgc, so.conv2d_same(x, kernels["rgc"], relu=True), RTOL, what="rgc_filter", bound=eb.conv(x, kernels["rgc"]))
    rgby = filters.rgby_filter(rgc)
    assert_close(rgby, so.conv2d_same(rgc, kernels["rgby"], relu=True), RTOL, what="rgby_filter", bound=eb.conv(rgc, kernels["rgby"]))
    orient = filters.orientation_filter(rgby)
    want = so.regulate(so.conv2d_same(rgby, kernels["stripe"], relu=True), kernels["blur"], 1.0, 0.1)
    stripe = so.conv2d_same(rgby, kernels["stripe"], relu=True)
    assert_close(orient, want, RTOL, what="orientation_filter",
                 bound=eb.regulate(stripe, kernels["blur"], 1.0, 0.1, eb.conv(rgby, kernels["stripe"])))
    le = apply_filter(orient, kernels["end"], relu=True, clip_hi=255.0)
    assert_close(le, so.conv2d_same(orient, kernels["end"], relu=True, clip_hi=255.0), RTOL, what="apply_filter",
                 bound=eb.conv(orient, kernels["end"]))
    assert rgc.dtype == np.float32 and rgc.shape == x.shape


def test_bad_arguments_raise_value_error(rt, kernels):
    x = np.zeros((1, 8, 8, 3), np.float32)
    with pytest.raises(ValueError):
        rt.conv2d_same(x, np.zeros((3, 3, 1, 3)))              # C_in mismatch
    with pytest.raises(ValueError):
        rt.conv2d_same(x[0], kernels["rgc"])                   # rank 3
    with pytest.raises(ValueError):
        rt.conv2d_same(x, np.zeros((17, 17, 3, 3)))            # outside the supported set
    with pytest.raises(ValueError):
        rt.regulate(x, kernels["blur"], 1.0, 0.1, flat_policy="bogus")
    with pytest.raises(ValueError):
        rt.gray_line_end(np.zeros((1, 8, 8, 1), np.float32), kernels["cs_gray"], np.zeros((3, 3, 1, 5)))


def test_bad_arguments_of_the_widened_entry_points(rt, kernels):
    """C ABI status codes of the section-8f entry points surface as ValueError / TypeError, never as a crash."""
    import ctypes as C
    from pysilent_amd import _lib
    from pysilent_amd.util import get_centroids
    from pysilent_amd.util.energy import get_boosting
    v1 = np.ones((1, 8, 8, 1), np.float32)
    v3 = np.ones((1, 8, 8, 3), np.float32)
    with pytest.raises(ValueError):
        get_centroids(v3, [1, 3, 3])                                   # needs a 1-channel value map
    with pytest.raises(ValueError):
        get_centroids(v1, [1, 0, 3])                                   # empty region
    with pytest.raises(TypeError):
        get_centroids([[1.0]], [1, 3, 3])                              # the reference's TypeError for foreign types
    with pytest.raises(ValueError):
        get_boosting(v1, np.ones((1, 4, 4, 1), np.float32))            # state geometry differs
    with pytest.raises(ValueError):
        rt.select_peaks(np.ones((1, 8, 8, 2), np.float32))             # 1 or 3 channels
    with pytest.raises(ValueError):
        rt.select_peaks(v3, 0.1, want=())                              # no output requested
    with pytest.raises(ValueError):
        rt.resize_nearest(v1, (0, 4))
    with pytest.raises(ValueError):
        rt.centroids(v1, -1, 2)
    lib, ctx = _lib.load(), rt.get_context(None)
    lev = (_lib.Extent * 1)(_lib.Extent(8, 8))
    assert lib.silent_affine_clip(ctx.handle, None, 4, None, None) == _lib.SILENT_E_INVALID
    assert lib.silent_boosting_step(ctx.handle, v1.ctypes.data, lev, 1, 1, None, v1.ctypes.data, v1.ctypes.data, None) \
        == _lib.SILENT_E_INVALID
    assert b"params" in lib.silent_last_error(ctx.handle)
    bad = _lib.BoostingParams(1.0, 1.0, 0, 10.0, 0.8, 0)               # recovery mode 0: "You must choose a type of recovery"
    out = np.empty_like(v1)
    assert lib.silent_boosting_step(ctx.handle, v1.ctypes.data, lev, 1, 1, C.byref(bad), v1.copy().ctypes.data,
                                    out.ctypes.data, None) == _lib.SILENT_E_INVALID
    assert b"recovery" in lib.silent_last_error(ctx.handle)


# ----------------------------------------------------------------------------- regulator

@pytest.mark.parametrize("policy", ["ieee", "zero"])
def test_regulate(rt, kernels, policy):
    x = so.conv2d_same(noise_frame(5, 40, 56, 3)[None], kernels["stripe"], relu=True)
    x[0, :18, :20] = 0          # an all-zero window: 0 * inf = NaN under "ieee"
    want = so.regulate(x, kernels["blur"], 1.0, 0.1, policy)
    got = rt.regulate(x, kernels["blur"], 1.0, 0.1, policy)
    assert np.isnan(want).any() == (policy == "ieee")
    assert_close(got, want, RTOL, what="regulate " + policy, bound=eb.regulate(x, kernels["blur"], 1.0, 0.1, None, policy))
    small = (x * np.float32(1e-3)).astype(np.float32)      # blur < 1: the pow branch is live
    assert_close(rt.regulate(small, kernels["blur"], 1.0, 0.5, policy),
                 so.regulate(small, kernels["blur"], 1.0, 0.5, policy), RTOL, what="regulate small",
                 bound=eb.regulate(small, kernels["blur"], 1.0, 0.5, None, policy))


def test_regulate_gray_blur(rt):
    from pysilent_amd.constant_convolutions import blur_tensor
    x = (noise_frame(6, 33, 47, 1)[None] / 255.0).astype(np.float32)
    for size in (3, 7, 5):
        b = blur_tensor(2, size, 1, 1)
        assert_close(rt.regulate(x, b, 2.0, 0.5), so.regulate(x, b, 2.0, 0.5), RTOL, what="blur %d" % size,
                     bound=eb.regulate(x, b, 2.0, 0.5))


# ----------------------------------------------------------------------------- fused gray pass

def ragged_pyramid(rt, seed, extents, c=1, n_frames=2):
    rng = np.random.default_rng(seed)
    levels = [rng.integers(0, 256, (n_frames, h, w, c)).astype(np.float32) for h, w in extents]
    return rt.PackedPyramid.from_levels(levels), levels


@pytest.mark.parametrize("K", [3, 4, 8])
def test_gray_line_end_fused(rt, kernels, K):
    extents = [(70, 131), (35, 66), (18, 33), (9, 17), (1, 1), (32, 64), (33, 65)]
    packed, levels = ragged_pyramid(rt, 20 + K, extents)
    bank = kernels["end%d" % K]
    cs, end = rt.gray_line_end(packed, kernels["cs_gray"], bank)
    for l, lev in enumerate(levels):
        want_cs = so.conv2d_same(lev, kernels["cs_gray"], relu=True)
        want_end = so.conv2d_same(want_cs, bank, relu=True, clip_hi=255.0)
        e_cs, e_end = eb.gray_chain(lev, kernels["cs_gray"], bank, want_cs)
        assert_close(cs.level(l), want_cs, RTOL, what="cs level %d" % l, bound=e_cs)
        # the second conv reads the GPU's own cs map: compare against the oracle applied to that map too
        gcs = np.ascontiguousarray(cs.level(l))
        assert_close(end.level(l), so.conv2d_same(gcs, bank, relu=True, clip_hi=255.0), RTOL, what="end|gpu-cs level %d" % l,
                     bound=eb.conv(gcs, bank))
        assert_close(end.level(l), want_end, RTOL, scale=255.0, what="end level %d" % l, bound=e_end)


def test_gray_line_end_equals_two_convs(rt, kernels):
    x = noise_frame(9, 45, 70, 1)[None]
    cs, end = rt.gray_line_end(x, kernels["cs_gray"], kernels["end4"], clip_hi=20.0)
    cs2 = rt.conv2d_same(x, kernels["cs_gray"], relu=True)
    end2 = rt.conv2d_same(cs2, kernels["end4"], relu=True, clip_hi=20.0)
    np.testing.assert_array_equal(cs, cs2)            # same fma order -> bit-identical
    np.testing.assert_array_equal(end, end2)
    only_cs, none = rt.gray_line_end(x, kernels["cs_gray"], kernels["end4"], want_end=False)
    assert none is None
    np.testing.assert_array_equal(only_cs, cs)


def test_config1_640x480_three_levels_center_surround(rt, kernels):
    """BASELINE config 1: one 640x480 gray frame, 3-level pyramid, center-surround only."""
    from pysilent_amd.util.zoom import classic_pyramid
    frame = noise_frame(0, 480, 640, 1)
    pyr = classic_pyramid(frame, 2.0, 3)
    assert pyr.extents == [(480, 640), (240, 320), (120, 160)]
    want_pyr = so.classic_pyramid(frame, 2.0, 3)
    cs, _ = rt.gray_line_end(pyr, kernels["cs_gray"], kernels["end4"], want_end=False)
    for l in range(3):
        assert_close(pyr.level(l), want_pyr[l], RTOL, what="pyramid config1 %d" % l, bound=eb.zoom(want_pyr[l]))
        assert_close(cs.level(l), so.conv2d_same(want_pyr[l], kernels["cs_gray"], relu=True), RTOL, scale=255.0,
                     what="cs config1 %d" % l, bound=eb.conv(want_pyr[l], kernels["cs_gray"], eb.zoom(want_pyr[l])))
    # the 3-channel kernel of the reference's own test on the frame replicated to 3 channels
    from pysilent_amd.constant_convolutions import center_surround_tensor
    k3 = center_surround_tensor(2, [0, 1, 0], [1, 0, 0], [0, 0, 1], [1, 0, 0])
    x3 = np.repeat(want_pyr[2], 3, axis=3)
    assert_close(rt.conv2d_same(x3, k3, relu=True), so.conv2d_same(x3, k3, relu=True), RTOL, what="cs 3ch", bound=eb.conv(x3, k3))


# ----------------------------------------------------------------------------- pyramid

@pytest.mark.parametrize("shape,center,scale", [((480, 640, 3), (288, 192), math.e ** .5),
                                                ((480, 640, 3), (160, 120), math.e ** .5),
                                                ((120, 160, 1), (40, 30), 2.0),
                                                ((97, 131, 3), (32, 24), 1.7)])
def test_from_image_reference_layout(rt, shape, center, scale):
    from pysilent_amd.util import zoom
    img = noise_frame(1, *shape)
    got = zoom.from_image(img, shape[2], center, scale)
    want = so.zoom_from_image(img, shape[2], center, scale)     # calls scipy.ndimage.zoom like the reference
    assert got.shape == want.shape and got.dtype == np.float32
    assert_close(got, want, RTOL, scale=255.0, what="from_image", bound=eb.zoom(want))


def test_from_image_against_the_reference_wrappers_own_outputs(rt, golden_pyramid):
    """zoom.from_image (HIP pyramid kernels) against pyramids produced by the reference's own image_to_zoom_tensor
    (tests/golden/pyramid.npz, generated by tests/golden/make_golden_pyramid.py): same level count and extents, values within
    the float32 tolerance (the GPU accumulates the 6 x 6 spline taps in float32 fmas, SciPy in float64)."""
    from pysilent_amd.util import zoom
    for name, case in golden_pyramid.items():
        if name.startswith("__"):
            continue
        img, want, par = case
        center, scale = [int(par[0]), int(par[1])], float(par[2])
        got = zoom.from_image(img, img.shape[2], center, scale)
        assert got.shape == want.shape and got.dtype == np.float32, name
        assert_close(got, want.astype(np.float32), RTOL, scale=255.0, what="from_image vs reference " + name, bound=eb.zoom(want))


def test_cast_interleave_and_device_frames_of_any_dtype(rt):
    """silent_cast_interleave: np.asarray(frame, float32) (recognition_testing.py:141) and the colour-plane slicing of
    from_image.py:54-64 as one strided cast kernel -- widening, plane extraction, interleaving; host and device forms; and
    the entry points fed GPU tensors that are not float32 (no torch kernel on that path)."""
    import ctypes as C
    import torch
    from pysilent_amd import _lib
    from pysilent_amd.util import zoom
    rng = np.random.default_rng(3)
    lib, ctx = _lib.load(), rt.get_context()
    for dt, code in ((np.uint8, _lib.DT_U8), (np.float64, _lib.DT_F64), (np.int32, _lib.DT_I32), (np.int16, _lib.DT_I16),
                     (np.uint16, _lib.DT_U16), (np.int64, _lib.DT_I64), (np.float32, _lib.DT_F32)):
        x = (rng.standard_normal((37, 5)) * 100).astype(dt) if dt in (np.float64, np.float32) else rng.integers(0, 120, (37, 5)).astype(dt)
        out = np.full((37, 7), -1.0, np.float32)
        ctx.check(lib.silent_cast_interleave(ctx.handle, x.ctypes.data, code, 37, 5, 1, 3, out.ctypes.data, 7, 2))
        want = np.full((37, 7), -1.0, np.float32)
        want[:, 2:5] = x[:, 1:4].astype(np.float32)
        np.testing.assert_array_equal(out, want, err_msg=str(dt))
    bad = np.zeros(4, np.float32)
    assert lib.silent_cast_interleave(ctx.handle, bad.ctypes.data, 99, 4, 1, 0, 1, bad.ctypes.data, 1, 0) == _lib.SILENT_E_UNSUPPORTED
    assert lib.silent_cast_interleave(ctx.handle, bad.ctypes.data, _lib.DT_F32, 4, 1, 1, 1, bad.ctypes.data, 1, 0) == _lib.SILENT_E_INVALID
    # device tensors of other dtypes through the wrappers
    img8 = rng.integers(0, 256, (60, 80, 3)).astype(np.uint8)
    a = zoom.from_image(torch.from_numpy(img8).cuda(), 3, (40, 30), 2.0)
    b = zoom.from_image(img8.astype(np.float32), 3, (40, 30), 2.0)
    np.testing.assert_array_equal(a.cpu().numpy(), b)
    np.testing.assert_array_equal(rt.nms3x3(torch.from_numpy(img8[None].astype(np.float64)).cuda(), "fired").cpu().numpy(),
                                  rt.nms3x3(img8[None].astype(np.float32), "fired"))
    # colour counts the kernels do not take interleaved (from_image.py zooms plane by plane): cut, zoom, interleave on the device
    for ncol in (2, 4):
        img = rng.integers(0, 256, (50, 70, ncol)).astype(np.float32)
        dev = zoom.from_image(torch.from_numpy(img).cuda(), ncol, (32, 24), 2.0)
        host = zoom.from_image(img, ncol, (32, 24), 2.0)
        assert dev.shape == host.shape and dev.shape[-1] == ncol
        np.testing.assert_array_equal(dev.cpu().numpy(), host)
        assert_close(host, so.zoom_from_image(img, ncol, (32, 24), 2.0), RTOL, scale=255.0, what="from_image %d colours" % ncol,
                     bound=eb.zoom(so.zoom_from_image(img, ncol, (32, 24), 2.0)))
    # layouts and dtypes outside the cast kernel's fast path (strided views, half precision, bool) are accepted like the
    # reference's np.asarray(frame, dtype=float32) accepts them: same result as the contiguous float32 copy
    f32 = torch.from_numpy(img8[None].astype(np.float32)).cuda()
    for view in (f32[:, :, ::2], f32.permute(0, 2, 1, 3), f32[..., :2], f32.half(), f32.bfloat16(), f32 > 128, f32.to(torch.int8)):
        want = rt.nms3x3(view.to(torch.float32).contiguous().cpu().numpy(), "fired")
        np.testing.assert_array_equal(rt.nms3x3(view, "fired").cpu().numpy(), want)


@pytest.mark.parametrize("shape,scale,n", [((135, 240, 1), 2.0, 5), ((135, 240, 3), 2.0, 4), ((270, 480, 1), math.e ** .5, 6),
                                           ((48, 64, 1), 2.0, 2), ((100, 37, 1), 1.3, 7)])
def test_classic_pyramid(rt, shape, scale, n):
    from pysilent_amd.util.zoom import classic_pyramid
    frames = np.stack([noise_frame(s, *shape) for s in range(2)])
    got = classic_pyramid(frames, scale, n)
    for f in range(2):
        want = so.classic_pyramid(frames[f], scale, n)
        for l in range(n):
            assert_close(got.level(l)[f:f + 1], want[l], RTOL, scale=255.0, what="frame %d level %d" % (f, l), bound=eb.zoom(want[l]))


@pytest.mark.parametrize("shape,scale,n", [((135, 240, 3), 2.0, 4), ((97, 131, 1), 1.7, 4), ((64, 64, 3), 2.0, 3), ((65, 129, 3), 2.0, 2),
                                           ((270, 480, 3), 2.0, 8), ((270, 480, 1), math.e ** .5, 6), ((33, 17, 3), 2.0, 2),
                                           ((100, 260, 3), 1.2, 3),
                                           ((270, 480, 1), 2 ** .5, 8), ((135, 240, 1), 1.5, 4), ((200, 300, 1), 1.4, 3),   # the dense slot layout
                                           ((200, 300, 1), 1.3, 3), ((200, 300, 1), 1.15, 3)])                              # 1.15: too dense for it
def test_pyramid_single_read_kernel_equals_unit_plus_region(rt, shape, scale, n):
    """silent_pyramid on classic pyramids: the single-read kernel (pyramid_stream_kernel, 1 and 3 channels) is
    bit-identical to the unit + region kernels (tuning knob PYRAMID = 1 selects those) and matches the oracle."""
    from pysilent_amd.util.zoom.from_image import classic_levels
    frames = np.stack([noise_frame(60 + s_, *shape) for s_ in range(3)])
    plan = rt.PyramidPlan(shape[0], shape[1], shape[2], classic_levels(shape[:2], scale, n))
    # (the host checks the row programs themselves: five rows of the first level in flight hold ratios down to about 1.3)
    assert plan.streamable == (scale >= 1.3 and shape[2] == 1)       # RGB plans keep unit + region kernels
    got = plan.run(frames)
    with rt.tuning(TUNE_PYRAMID, 1):
        two = plan.run(frames)
    np.testing.assert_array_equal(got.data, two.data)
    want = so.classic_pyramid(frames[2], scale, n)
    for l in range(n):
        assert_close(got.level(l)[2:3], want[l], RTOL, scale=255.0, what="level %d" % l, bound=eb.zoom(want[l]))


def test_pyramid_known_answers(rt):
    from pysilent_amd.util.zoom import classic_pyramid
    img = np.zeros((11, 11, 1), np.float32)
    img[5, 5, 0] = 120.0 * 120.0
    lvl0 = classic_pyramid(img, 2.0, 1).level(0)[0, :, :, 0]
    np.testing.assert_allclose(lvl0[5, 3:8], [66., 1716., 4356., 1716., 66.], rtol=1e-6)   # [1,26,66,26,1]^2 row
    const = np.full((41, 57, 1), 77.0, np.float32)      # mirror taps: a constant image stays constant
    for lev, want in zip(classic_pyramid(const, 2.0, 3).levels(), so.classic_pyramid(const, 2.0, 3)):
        assert (want == 77.0).all()                      # (sizes chosen so that scipy's last-row rule stays out)
        np.testing.assert_allclose(lev, 77.0, rtol=1e-6)
    # scipy's mode='constant' rule: 23 * (47/23) lands one ulp above 47 -> the last output row is cval = 0
    img = noise_frame(2, 48, 64, 1)
    lvl1 = classic_pyramid(img, 2.0, 2).level(1)[0, :, :, 0]
    assert (lvl1[-1] == 0).all() and (lvl1[:-1] != 0).any()


# ----------------------------------------------------------------------------- pointwise, nms, selection

def test_pad_value_nms_bit_exact(rt):
    x = noise_frame(4, 29, 43, 3)[None]
    x[0, 3:9, 4:12] = 0                         # a zero plateau: every pixel of it "fires"
    pads = [[0, 0], [2, 2], [2, 2], [0, 0]]
    from pysilent_amd.util.selection import pad_inwards
    from pysilent_amd.util.color import get_value_from_color
    from pysilent_amd.util.energy import has_fired, local_maxima
    np.testing.assert_array_equal(pad_inwards(x, pads), so.pad_inwards(x, pads))
    np.testing.assert_array_equal(pad_inwards(x, [[0, 0], [1, 3], [0, 5], [0, 0]]),
                                  so.pad_inwards(x, [[0, 0], [1, 3], [0, 5], [0, 0]]))
    for big in ([[0, 0], [0, 30], [0, 0], [0, 0]], [[0, 0], [15, 15], [1, 1], [0, 0]], [[0, 0], [0, 0], [40, 3], [0, 0]]):
        assert not pad_inwards(x, big).any()                # paddings that use up an axis leave nothing (as the oracle)
        np.testing.assert_array_equal(pad_inwards(x, big), so.pad_inwards(x, big))
    np.testing.assert_array_equal(get_value_from_color(x), so.value_from_color(x))
    np.testing.assert_array_equal(local_maxima(x), so.nms3x3(x, "product"))
    np.testing.assert_array_equal(has_fired(x), so.nms3x3(x, "fired"))
    v = so.value_from_color(x)
    np.testing.assert_array_equal(has_fired(v), so.nms3x3(v, "fired"))


@pytest.mark.parametrize("p", [0.1, 0.5, 0.0, 1.0, 0.37])
def test_top_value_points_bit_exact(rt, p):
    from pysilent_amd.util.selection import top_value_points
    x = np.stack([noise_frame(s, 31, 45, 3) for s in (0, 1, 2)])
    x[1] *= 0.25
    np.testing.assert_array_equal(top_value_points(x, p), so.top_value_points(x, p))
    v = (so.value_from_color(x) - np.float32(100)).astype(np.float32)           # negative values too
    np.testing.assert_array_equal(top_value_points(x, p, v), so.top_value_points(x, p, v))


@pytest.mark.parametrize("shape,region", [((2, 192, 288, 3), (1, 96, 144, 3)), ((3, 37, 53, 3), (1, 18, 26, 3)),
                                          ((1, 37, 53, 3), (1, 37, 53, 3)), ((2, 40, 40, 3), (1, 10, 14, 3)),
                                          ((1, 19, 27, 1), (1, 7, 9, 1))])
def test_max_value_indices_region_bit_exact(rt, shape, region):
    from pysilent_amd.util.selection import max_value_indices_region
    x = np.random.default_rng(8).integers(0, 256, shape).astype(np.float32)
    x[0, :shape[1] // 2, :shape[2] // 2] = 0     # an all-zero quadrant: every pixel of it is emitted
    got = max_value_indices_region(x, region)
    want = so.max_value_indices_region(x, region)
    assert got.dtype == np.int64
    np.testing.assert_array_equal(got, want)
    v = so.value_from_color(x)
    np.testing.assert_array_equal(max_value_indices_region(x, region, v), co.max_value_indices_region(x, region, v))


def test_max_value_indices_capacity_error(rt):
    v = np.zeros((1, 16, 16, 1), np.float32)
    idx, counts = rt.max_value_indices_region(v, [(8, 8)])
    assert counts[0] == 256
    with pytest.raises(ValueError, match="cap_per_frame"):
        rt.max_value_indices_region(v, [(8, 8)], cap_per_frame=10)


# ----------------------------------------------------------------------------- centroids

@pytest.mark.parametrize("shape,region", [((2, 192, 288, 1), [1, 3, 3]), ((1, 37, 53, 1), [1, 3, 3]),
                                          ((3, 20, 31, 1), [1, 4, 5]), ((1, 7, 7, 1), [1, 2, 2])])
def test_get_centroids(rt, shape, region):
    from pysilent_amd.util import get_centroids
    rng = np.random.default_rng(12)
    v = (rng.random(shape) * (rng.random(shape) > 0.6)).astype(np.float32)      # sparse: some cells are empty -> NaN
    v[0, :6, :9] = 0
    dist, total = get_centroids(v, region)
    wd, wt = so.get_centroids(v, region)
    assert np.isnan(wd).any()
    assert dist.shape == wd.shape and total.shape == wt.shape
    assert_close(total, wt, RTOL, what="total_pool")
    assert_close(dist, wd, RTOL, scale=float(max(shape[1], shape[2])), what="value_centroids")


def test_get_centroids_known_answer_and_packed(rt):
    from pysilent_amd.util import get_centroids
    v = np.zeros((1, 6, 9, 1), np.float32)
    v[0, 1, 4, 0] = 2.0
    v[0, 2, 5, 0] = 2.0                                   # cell (0,1): centroid (x, y) = (4.5, 1.5), total 4
    dist, total = get_centroids(v, [1, 3, 3])
    assert total[0, 0, 1, 0] == 4.0 and total.sum() == 4.0
    np.testing.assert_array_equal(dist[0, :3, 3:6, 0], [[3, 2, 2], [2, 1, 1], [2, 1, 1]])
    assert np.isnan(dist[0, 0, 0, 0])                     # empty cell: 0/0 like the reference
    packed, levels = ragged_pyramid(rt, 5, [(12, 20), (7, 9)], c=1, n_frames=2)
    d, t = get_centroids(packed, [1, 3, 3])
    for l, lev in enumerate(levels):
        wd, wt = so.get_centroids(lev, [1, 3, 3])
        assert_close(d.level(l), wd, RTOL, scale=20.0, what="packed dist %d" % l)
        assert_close(t.level(l), wt, RTOL, what="packed total %d" % l)


# ----------------------------------------------------------------------------- boosting state

@pytest.mark.parametrize("inp,const", [(False, True), (True, False), (True, True)])
@pytest.mark.parametrize("vis", [False, True])
def test_get_boosting_sequence(rt, inp, const, vis):
    """Five successive frames of one stream: fired masks bit-exact, state within tolerance at every step."""
    from pysilent_amd.util.energy import get_boosting, initialize_boosting
    rng = np.random.default_rng(31)
    shape = (3, 41, 67, 1)
    state = initialize_boosting(np.empty(shape, np.float32))
    assert state.dtype == np.float32 and (state == 8).all()
    want_state = so.initialize_boosting(np.empty(shape))
    for step in range(5):
        x = np.floor(rng.random(shape) * 256).astype(np.float32) * (rng.random(shape) > 0.3)
        got = get_boosting(x, state, 1, 1, inp, const, for_visualizing=vis)
        want = so.get_boosting(x, want_state, 1, 1, inp, const, for_visualizing=vis)
        want_state = want[-1] if vis else want[1]
        np.testing.assert_array_equal(got[0], want[0], err_msg="fired, step %d" % step)
        assert_close(got[1], want[1], RTOL, what="energy map, step %d" % step)
        assert_close(state, want_state, RTOL, what="state, step %d" % step)
    assert state.min() >= -1 and state.max() <= 1


def test_get_boosting_first_step_from_initial_state_and_packed(rt):
    """From the initial state 8 (boosting.py:6-7), packed ragged levels, wider clip range, device-resident state."""
    import torch
    from pysilent_amd.util.energy import get_boosting, initialize_boosting
    packed, levels = ragged_pyramid(rt, 9, [(24, 40), (11, 13), (5, 3)], c=1, n_frames=2)
    state = initialize_boosting(packed)
    fired, energy = get_boosting(packed, state, exhaustion_max=2, excitation_max=8)
    for l, lev in enumerate(levels):
        wf, we = so.get_boosting(lev, so.initialize_boosting(lev), 2, 8)
        np.testing.assert_array_equal(fired.level(l), wf)
        assert_close(energy.level(l), we, RTOL, what="energy %d" % l)
        assert_close(state.level(l), we, RTOL, what="state %d" % l)
    x = torch.rand((1, 16, 16, 1), device="cuda") * 255
    st = initialize_boosting(x)
    f, e = get_boosting(x, st)
    wf, we = so.get_boosting(x.cpu().numpy(), so.initialize_boosting(np.empty((1, 16, 16, 1))))
    np.testing.assert_array_equal(f.cpu().numpy(), wf)
    assert_close(st.cpu().numpy(), we, RTOL, what="device state")
    with pytest.raises(ValueError, match="type of recovery"):
        get_boosting(x, st, input_based_recovery=False, constant_recovery=False)
    with pytest.raises(ValueError, match="in place"):
        get_boosting(np.ones((1, 4, 4, 1), np.float32), np.ones((1, 4, 4, 1), np.float64))


# ----------------------------------------------------------------------------- display graph

def test_get_bw_from_color(rt):
    from pysilent_amd.util.color import get_bw_from_color
    rng = np.random.default_rng(3)
    x = (np.floor(rng.random((2, 21, 37, 3)) * 3) - 1).astype(np.float32)       # -1 / 0 / 1: many cancelling sums
    x[0, 0, 0] = [np.nan, 0, 0]
    x[0, 0, 1] = [np.inf, -np.inf, 0]
    x[0, 0, 2] = [1e-30, 0, 0]
    got = get_bw_from_color(x)
    want = so.bw_from_color(x)
    assert 0.2 < want.mean() < 0.95
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(get_bw_from_color(x[..., :1]), so.bw_from_color(x[..., :1]))
    packed, levels = ragged_pyramid(rt, 9, [(12, 20), (7, 9)], c=3, n_frames=2)
    gp = get_bw_from_color(packed)
    for l in range(2):
        np.testing.assert_array_equal(gp.level(l), so.bw_from_color(levels[l]))


def test_to_channels(rt):
    """util/color/to_channels.py:6-16: tile the single channel; other channel counts cannot satisfy its set_shape."""
    from pysilent_amd.util.color import to_channels
    x = noise_frame(8, 19, 23, 1)[None]
    x[0, 0, 0, 0] = np.nan
    x[0, 0, 1, 0] = np.inf
    np.testing.assert_array_equal(to_channels(x), np.tile(x, (1, 1, 1, 3)))
    np.testing.assert_array_equal(to_channels(x, 2), np.tile(x, (1, 1, 1, 2)))
    packed, levels = ragged_pyramid(rt, 5, [(12, 20), (7, 9)], c=1, n_frames=2)
    got = to_channels(packed, 4)
    for l in range(2):
        np.testing.assert_array_equal(got.level(l), np.tile(levels[l], (1, 1, 1, 4)))
    with pytest.raises(ValueError):
        to_channels(noise_frame(8, 5, 5, 3)[None])


def test_affine_clip_and_resize_nearest(rt):
    rng = np.random.default_rng(77)
    x = (rng.standard_normal((2, 13, 17, 3)) * 100).astype(np.float32)
    x[0, 0, 0, 0] = np.nan
    for kw in (dict(div=255.0), dict(mul=255 / 4.0, lo=1.0, hi=256.0, post_add=-1.0), dict(mul=-255.0, add=255.0)):
        got = rt.affine_clip(x, **kw)
        np.testing.assert_array_equal(got, so.affine_clip(x, **kw))
    for out in ((7, 10), (13, 17), (30, 19), (1, 1)):
        np.testing.assert_array_equal(rt.resize_nearest(x, out), so.resize_nearest_tf1(x, *out))
    packed, levels = ragged_pyramid(rt, 3, [(12, 20), (7, 9)], c=1, n_frames=2)
    got = rt.resize_nearest(packed, [(7, 12), (9, 4)])
    np.testing.assert_array_equal(got.level(0), so.resize_nearest_tf1(levels[0], 7, 12))
    np.testing.assert_array_equal(got.level(1), so.resize_nearest_tf1(levels[1], 9, 4))


def test_line_end_displayer_three_frames(rt, kernels):
    """The reference application graph (recognition_testing.py:60-144) over three successive frames of one
    stream: all six fetched tensors and the boosting state against the oracle."""
    from pysilent_amd.recognition_testing import LineEndDisplayer
    from pysilent_amd.util import zoom
    disp = LineEndDisplayer(output_size=(96, 64))
    ks = {k: kernels[k] for k in ("rgc", "rgby", "stripe", "blur", "end")}
    want_state = None
    names = ["orient", "255 - centroids * 255", "255 - centroids2 * 255", "fired * 255", "update", "padded"]
    for step in range(3):
        frame = structured_frame(40 + step, 150, 230, 3)
        res = disp.callback(frame)
        assert len(res) == 7 and res[0] is frame
        pyr = zoom.from_image(frame.astype(np.float32), 3, (96, 64), disp.zoom_ratio)
        assert pyr.shape[1:] == (64, 96, 3) and len(res[1]) == pyr.shape[0]
        if want_state is None:
            want_state = np.full((pyr.shape[0], 22, 32, 1), 8, np.float32)
        # the filter chain against the oracle chain; everything after pad_inwards against the oracle applied to the
        # GPU's own padded map (a centroid is a ratio of sums: where a cell holds only rounding residue the ratio
        # is ill-conditioned, so stage-wise comparison is the meaningful one -- same idea as "end|gpu-cs" above)
        full, _ = so.line_end_displayer_run(pyr, want_state, ks)
        got = [np.stack(r) for r in res[1:]]
        assert_close(got[0], full[0], RTOL, what="orient, frame %d" % step)
        assert_close(got[5], full[5], RTOL, scale=255.0, what="padded, frame %d" % step)
        tail, want_state = so.line_end_displayer_tail(got[5], want_state)
        for i, name in enumerate(names[1:5]):
            assert got[1 + i].shape == tail[i].shape, name
            # 255 - |c - x| * 255: the terms are pixel coordinates (< h + w) times 255
            scale = 255.0 * (64 + 96) if i < 2 else 255.0
            assert_close(got[1 + i], tail[i], RTOL, scale=scale, what="%s, frame %d" % (name, step))
        assert_close(disp.get_state(), want_state, RTOL, what="state, frame %d" % step)
    # (above: the default, native path -- one library call per frame, silent_displayer_step.)  The per-op path gives the same bits,
    # over six frames: the native displayer's first frame runs eagerly, the next two capture one graph per result slot, the rest replay
    assert disp.native and disp._native is not None
    native, per_op = LineEndDisplayer(output_size=(96, 64)), LineEndDisplayer(output_size=(96, 64), native=False)
    held = []
    for step in range(6):
        frame = structured_frame(60 + step, 150, 230, 3)
        a, b = native.callback(frame, copy=False), per_op.callback(frame)
        for i in range(1, 7):
            np.testing.assert_array_equal(np.stack(a[i]), np.stack(b[i]), err_msg="native vs per-op, frame %d, output %d" % (step, i))
        np.testing.assert_array_equal(native.get_state(), per_op.get_state())
        held.append((np.stack(a[1]).copy(), a[1]))
        if step >= 1:       # the previous frame's zero-copy views are still intact (two result slots alternate)
            np.testing.assert_array_equal(held[step - 1][0], np.stack(held[step - 1][1]))
    st = native.get_state()
    native.set_state(st * 0 + 3.0)
    per_op.set_state(st * 0 + 3.0)
    frame = structured_frame(70, 150, 230, 3)
    a, b = native.callback(frame), per_op.callback(frame)
    for i in range(1, 7):
        np.testing.assert_array_equal(np.stack(a[i]), np.stack(b[i]))
    # the same three frames through a captured HIP graph (per-op path): bit-identical outputs and state
    eager = LineEndDisplayer(output_size=(96, 64), native=False)
    graphed = LineEndDisplayer(output_size=(96, 64), native=False, use_graph=True)
    for step in range(3):
        frame = structured_frame(40 + step, 150, 230, 3)
        a, b = eager.callback(frame), graphed.callback(frame)
        for i in range(1, 7):
            np.testing.assert_array_equal(np.stack(a[i]), np.stack(b[i]))
        np.testing.assert_array_equal(eager.get_state(), graphed.get_state())
    # camera frames are uint8: they cross PCIe as bytes and are widened on the device -- same maps as the float path
    cam = np.clip(structured_frame(43, 150, 230, 3), 0, 255).astype(np.uint8)
    a = LineEndDisplayer(output_size=(96, 64)).callback(cam)
    b = LineEndDisplayer(output_size=(96, 64)).callback(cam.astype(np.float32))
    assert a[0] is cam
    for i in range(1, 7):
        np.testing.assert_array_equal(np.stack(a[i]), np.stack(b[i]))
    saved = disp.get_state()
    disp.set_state(saved * 0 + 8)
    assert (disp.get_state() == 8).all()
    shown = disp.display(structured_frame(1, 150, 230, 3))
    assert len(shown) == 7 and float(np.nanmax(shown[6])) <= 1.0
    with pytest.raises(NotImplementedError):
        disp.run_camera()


@pytest.mark.parametrize("shape", [(480, 640), (1080, 1920), (333, 517)])
def test_line_end_displayer_camera_geometry_native_equals_per_op(rt, shape):
    """The application graph at the reference's own geometry (camera frames, output_size (288, 192), zoom e ** .5 -- what bench.py's
    latency record times): the native displayer reads only the rectangle of the uint8 frame its pyramid needs, straight from pinned
    host memory, and its kernels write the pinned result slot themselves -- every fetched tensor and the state bit-identical to the
    per-op path over four different frames (a pixel used from outside that rectangle, or a result not yet landed, would show)."""
    from pysilent_amd.recognition_testing import LineEndDisplayer
    h, w = shape
    native, per_op = LineEndDisplayer(), LineEndDisplayer(native=False)
    rng = np.random.default_rng(5)
    for step in range(4):
        frame = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        if step == 2:
            frame[:] = structured_frame(7, h, w, 3).clip(0, 255).astype(np.uint8)
        a, b = native.callback(frame), per_op.callback(frame)
        for i in range(1, 7):
            np.testing.assert_array_equal(np.stack(a[i]), np.stack(b[i]), err_msg="frame %d, output %d" % (step, i))
        np.testing.assert_array_equal(native.get_state(), per_op.get_state())


@pytest.mark.parametrize("seed", range(10))
def test_line_end_displayer_random_geometries_native_equals_per_op(rt, seed):
    """Random camera geometries through the native application graph (zero copy at both ends, only the crop union of the frame is
    converted, tiny-launch tile heights / segment heights) against the per-op path: frame sizes, output sizes, zoom ratios and frame
    dtypes drawn at random; every fetched tensor and the state bit for bit over three frames, raw views and held arrays alike."""
    from pysilent_amd.recognition_testing import LineEndDisplayer
    rng = np.random.default_rng(1000 + seed)
    ow, oh = int(rng.integers(12, 80)) * 4, int(rng.integers(24, 200))
    ratio = float(rng.choice([1.3, 2 ** .5, 1.5, np.e ** .5, 2.0]))
    h = int(oh * ratio ** rng.uniform(0.6, 2.6)) + int(rng.integers(1, 9))
    w = (int(ow * ratio ** rng.uniform(0.6, 2.6)) + int(rng.integers(1, 9)) + 3) // 4 * 4
    dtype = [np.uint8, np.uint8, np.float32, np.uint16, np.float64][int(rng.integers(0, 5))]
    kw = dict(output_size=(ow, oh), zoom_ratio=ratio)
    native, per_op = LineEndDisplayer(**kw), LineEndDisplayer(native=False, **kw)
    for step in range(3):
        frame = rng.integers(0, 256, (h, w, 3)).astype(dtype)
        a = native.callback(frame, copy=bool(step & 1))
        b = per_op.callback(frame)
        assert native._native is not None, "the native path must take camera frames of this dtype"
        for i in range(1, 7):
            np.testing.assert_array_equal(np.stack(a[i]), np.stack(b[i]),
                                          err_msg="%dx%d %s -> %dx%d ratio %.3f, frame %d, output %d" % (w, h, np.dtype(dtype).name, ow, oh, ratio, step, i))
        np.testing.assert_array_equal(native.get_state(), per_op.get_state())


def test_line_end_displayer_results_outlive_frames_shape_changes_and_close(rt):
    """like the reference's session.run (recognition_testing.py:132-144) ``callback`` returns FRESH arrays -- a consumer
    (the reference's asynchronous display loop) may keep them over any number of frames; the zero-copy views (``copy=False``,
    ``FrameDisplayer.step``, ``frame_buffer``) stay readable over a frame-shape change (the old displayer is dropped, not
    destroyed) and over an explicit ``close()`` (deferred while such views are alive)."""
    import gc
    from pysilent_amd.recognition_testing import LineEndDisplayer
    disp = LineEndDisplayer(output_size=(96, 64))
    f0 = structured_frame(80, 150, 230, 3)
    kept = disp.callback(f0)                                   # the caller's for as long as it holds them (a pinned slot of their own)
    want = [np.stack(kept[i]).copy() for i in range(1, 7)]
    views = disp.callback(f0 * 0 + 7.0, copy=False)            # views of a pinned slot (another state step: other values)
    want_views = [np.stack(views[i]).copy() for i in range(1, 7)]
    old = disp._native[1]
    fb = old.frame_buffer
    for step in range(5):                                      # five more frames: both slots overwritten
        disp.callback(structured_frame(81 + step, 150, 230, 3))
    for i in range(6):
        np.testing.assert_array_equal(np.stack(kept[1 + i]), want[i])
    # a new frame shape: a new displayer; the old one's pinned memory stays under the views we hold
    views = disp.callback(f0 * 0 + 7.0, copy=False)
    want_views = [np.stack(views[i]).copy() for i in range(1, 7)]
    fb[...] = 5.0                                              # (the old displayer's pinned input buffer: nobody writes it from here on)
    disp.callback(structured_frame(90, 170, 250, 3))
    assert disp._native[1] is not old
    gc.collect()
    for i in range(6):
        np.testing.assert_array_equal(np.stack(views[1 + i]), want_views[i])
    assert float(fb[0, 0, 0]) == 5.0
    # an explicit close with views outstanding is deferred; stepping a closed displayer raises
    old.close()
    with pytest.raises(RuntimeError):
        old.step(f0)
    for i in range(6):
        np.testing.assert_array_equal(np.stack(views[1 + i]), want_views[i])
    assert float(fb[0, 0, 0]) == 5.0
    del views, fb, old
    gc.collect()                                               # now it is destroyed (nothing to assert but "no crash")
    fresh = LineEndDisplayer(output_size=(96, 64))
    again = fresh.callback(f0)
    for i in range(6):
        np.testing.assert_array_equal(np.stack(again[1 + i]), want[i])
    # the slot pool: results that are dropped give their slot back (a long run needs two or three slots, not one per frame) ...
    fd = fresh._native[1]
    for step in range(20):
        r = fresh.callback(structured_frame(100 + step, 150, 230, 3))
    assert 1 <= len(fd._held) <= 3
    # ... and a consumer that keeps EVERYTHING gets copies once MAX_HELD_SLOTS slots are taken: still every frame's own values
    per_op = LineEndDisplayer(output_size=(96, 64), native=False)
    per_op.callback(f0)                                        # (compiles the pyramid shape; the state is set next)
    per_op.set_state(fresh.get_state())
    hoard, wants = [], []
    for step in range(fd.MAX_HELD_SLOTS + 4):
        fr = structured_frame(200 + step, 150, 230, 3)
        hoard.append(fresh.callback(fr))
        wants.append([np.stack(x) for x in per_op.callback(fr)[1:]])
    assert len(fd._held) == fd.MAX_HELD_SLOTS
    for got, want_f in zip(hoard, wants):
        for i in range(6):
            np.testing.assert_array_equal(np.stack(got[1 + i]), want_f[i])


# ----------------------------------------------------------------------------- RGB chain

@pytest.mark.parametrize("policy,frame", [("ieee", "noise"), ("zero", "structured"), ("ieee", "structured")])
def test_rgb_chain(rt, kernels, policy, frame):
    img = noise_frame(2, 64, 96, 3) if frame == "noise" else structured_frame(2, 64, 96, 3, 30)
    x = img[None]
    want = so.rgb_line_end_chain(x, kernels, policy)
    got = rt.rgb_line_end(x, kernels, flat_policy=policy)
    if frame == "noise":
        assert not np.isnan(want["line_end"]).any()            # SURVEY hard part 3: noise frames stay finite
    bound = eb.rgb_chain(x, kernels, want, policy)
    assert_close(got["orient"], want["orient"], RTOL, what="orient", bound=bound["orient"])
    # later stages: compare against the oracle continued from the GPU's own orient (isolates each stage) ...
    g_orient = np.ascontiguousarray(got["orient"])
    le = so.pad_inwards(so.conv2d_same(g_orient, kernels["end"], relu=True, clip_hi=255.0), [[0, 0], [2, 2], [2, 2], [0, 0]])
    assert_close(got["line_end"], le, RTOL, what="line_end|gpu-orient", bound=eb.pad(eb.conv(g_orient, kernels["end"]), 2))
    np.testing.assert_array_equal(got["value"], so.value_from_color(np.ascontiguousarray(got["line_end"])))
    # ... and end to end
    assert_close(got["line_end"], want["padded"], RTOL, scale=255.0, what="line_end", bound=bound["padded"])
    assert_close(got["value"], want["value"], RTOL, scale=255.0, what="value", bound=bound["value"])


def test_rgb_chain_structured_and_dense_kernels_agree(rt, kernels):
    """The specialised chain kernel (diagonal rgc, channel-sum stripe, two-group rgby / end: 189 fmas per pixel) against
    the dense one (tuning knob RGB = 1: 373 fmas) and against the one without the two-group forms (2): re-association
    only, well inside the 1e-5 budget."""
    frames = np.stack([noise_frame(90 + i, 70, 131, 3) for i in range(2)])
    fast = rt.rgb_line_end(frames, kernels)
    with rt.tuning(TUNE_RGB, 1):
        dense = rt.rgb_line_end(frames, kernels)
    with rt.tuning(TUNE_RGB, 2):
        mid = rt.rgb_line_end(frames, kernels)
    with rt.tuning(TUNE_RGB, 8):                     # 90-row tiles although the launch is small (default here: 18 rows)
        tall = rt.rgb_line_end(frames, kernels)
    with rt.tuning(TUNE_RGB, 9):
        tall_dense = rt.rgb_line_end(frames, kernels)
    for name in ("orient", "line_end", "value"):
        np.testing.assert_array_equal(fast[name], tall[name])           # the tile height does not change a bit
        np.testing.assert_array_equal(dense[name], tall_dense[name])
    for name in ("orient", "line_end", "value"):
        # two float32 evaluation orders against each other: range-relative only (rel_floor=None)
        assert_close(fast[name], dense[name], 2e-6, scale=255.0, what=name + " structured vs dense", rel_floor=None)
        assert_close(mid[name], dense[name], 2e-6, scale=255.0, what=name + " basic vs dense", rel_floor=None)
        assert not np.array_equal(fast[name], dense[name]) or name == "value"      # really different code paths
    want = so.rgb_line_end_chain(frames, kernels)
    bound = eb.rgb_chain(frames, kernels, want)
    assert_close(fast["line_end"], want["padded"], RTOL, scale=255.0, what="line_end vs oracle", bound=bound["padded"])
    assert_close(fast["orient"], want["orient"], RTOL, what="orient vs oracle", bound=bound["orient"])
    assert_close(dense["line_end"], want["padded"], RTOL, scale=255.0, what="dense line_end vs oracle", bound=bound["padded"])


@pytest.mark.parametrize("shape", [(2, 70, 131, 3), (1, 33, 448, 3), (1, 211, 449, 3), (3, 19, 5, 3), (1, 1, 1, 3),
                                   (1, 100, 113, 3), (1, 95, 912, 3)])
@pytest.mark.parametrize("variant", [64, 1, 2])
def test_rgb_pair_kernel_is_bit_identical_to_the_one_pixel_kernel(rt, kernels, shape, variant):
    """rgb_line_end2_kernel (two adjacent pixels per lane, v_pk_fma_f32 with the SGPR weight for both halves) against
    rgb_line_end_kernel (TUNE_RGB bit 16): each half of a packed fma chain is the fmaf chain of its pixel in the same order,
    so every output bit must agree -- for the two-group (64: the symmetric forms off), the basic and the dense instantiation, on
    odd widths, widths around the 112 / 448-column wave / tile boundaries, single pixels, NaN / inf pixels, both flat policies
    and 18 / 90-row tiles.  (The symmetric forms -- the default on the reference's kernels -- sum a pixel's left and right taps in
    an order that depends on its column parity: test_rgb_symmetric_forms_against_the_two_group_kernel.)"""
    rng = np.random.default_rng(shape[1] * 1000 + shape[2])
    frames = np.stack([noise_frame(50 + i, shape[1], shape[2], 3) for i in range(shape[0])])
    if shape[1] > 8:
        frames[0, 3:9, : max(1, shape[2] // 3)] = 0.0            # a flat region: 0 * inf under 'ieee'
    if shape[1] * shape[2] > 64:
        ys = rng.integers(0, shape[1], 5)
        xs = rng.integers(0, shape[2], 5)
        frames[0, ys[0], xs[0], 1] = np.nan
        frames[0, ys[1], xs[1], 0] = -np.nan
        frames[0, ys[2], xs[2], 2] = np.inf
        frames[0, ys[3], xs[3], 0] = -np.inf
    for policy in ("ieee", "zero"):
        for tall in (0, 8):
            with rt.tuning(TUNE_RGB, variant | tall):
                pair = rt.rgb_line_end(frames, kernels, flat_policy=policy)
            with rt.tuning(TUNE_RGB, variant | tall | 16):
                one = rt.rgb_line_end(frames, kernels, flat_policy=policy)
            for name in ("orient", "line_end", "value"):
                a, b = pair[name], one[name]
                assert a.shape == b.shape
                assert np.array_equal(np.isnan(a), np.isnan(b)), (name, policy, tall)
                np.testing.assert_array_equal(np.nan_to_num(a, nan=7.0), np.nan_to_num(b, nan=7.0), err_msg="%s %s %d" % (name, policy, tall))


@pytest.mark.parametrize("shape", [(2, 70, 131, 3), (1, 33, 448, 3), (1, 211, 449, 3), (3, 19, 5, 3), (1, 1, 1, 3), (1, 2, 2, 3),
                                   (1, 100, 113, 3), (1, 95, 912, 3)])
def test_rgb_symmetric_forms_against_the_two_group_kernel(rt, kernels, shape):
    """The default instantiation on the reference's kernels (csrc/silent_rgb2.h, SYM: rgc folded over both mirror axes, rgby as
    channel mix -> one symmetric profile -> centre mix, left / right taps as swapped-half packed fmas without moves, relu + clip
    of the line-end as v_maximum3 / v_minimum3) against the two-group instantiation (TUNE_RGB bit 6), which is bit-identical to
    the one-pixel kernel.  Re-association only: the NaN / inf footprint must be THE SAME (zero weights of the channel mixes are
    multiplied, not skipped, like the reference's 0 * x), finite values agree to a few ulp of the response range and both sit
    inside the oracle's element-wise rounding bound."""
    rng = np.random.default_rng(shape[1] * 1000 + shape[2])
    frames = np.stack([noise_frame(150 + i, shape[1], shape[2], 3) for i in range(shape[0])])
    if shape[1] > 8:
        frames[0, 3:9, : max(1, shape[2] // 3)] = 0.0            # a flat region: 0 * inf under 'ieee'
    clean = frames.copy()
    if shape[1] * shape[2] > 64:
        ys = rng.integers(0, shape[1], 5)
        xs = rng.integers(0, shape[2], 5)
        frames[0, ys[0], xs[0], 1] = np.nan
        frames[0, ys[1], xs[1], 0] = -np.nan
        frames[0, ys[2], xs[2], 2] = np.inf
        frames[0, ys[3], xs[3], 0] = -np.inf
    for policy in ("ieee", "zero"):
        for x in (frames, clean):
            for tall in (0, 8):
                with rt.tuning(TUNE_RGB, tall):
                    sym = rt.rgb_line_end(x, kernels, flat_policy=policy)
                with rt.tuning(TUNE_RGB, tall | 64):
                    two = rt.rgb_line_end(x, kernels, flat_policy=policy)
                for name in ("orient", "line_end", "value"):
                    a, b = sym[name], two[name]
                    assert np.array_equal(np.isnan(a), np.isnan(b)), (name, policy, tall)
                    assert np.array_equal(np.isinf(a), np.isinf(b)), (name, policy, tall)
                    fin = np.isfinite(b)
                    np.testing.assert_array_equal(np.signbit(a[~fin & ~np.isnan(b)]), np.signbit(b[~fin & ~np.isnan(b)]))
                    if fin.any() and x is clean:
                        assert_close(np.where(fin, a, 0), np.where(fin, b, 0), 2e-6, scale=255.0, what="%s %s symmetric vs two-group" % (name, policy), rel_floor=None)
        want = so.rgb_line_end_chain(clean, kernels, policy)
        bound = eb.rgb_chain(clean, kernels, want, policy)
        got = rt.rgb_line_end(clean, kernels, flat_policy=policy)
        assert_close(got["orient"], want["orient"], RTOL, what="orient " + policy, bound=bound["orient"])
        assert_close(got["line_end"], want["padded"], RTOL, scale=255.0, what="line_end " + policy, bound=bound["padded"])
        assert_close(got["value"], want["value"], RTOL, scale=255.0, what="value " + policy, bound=bound["value"])


@pytest.mark.parametrize("knob", [0, 64, 2, 1, 64 | 16, 2 | 16])
@pytest.mark.parametrize("shape", [(2, 70, 131, 3), (1, 33, 448, 3), (1, 19, 5, 3), (1, 1, 1, 3), (1, 2, 3, 3), (1, 100, 113, 3)])
def test_rgb_chain_nonfinite_pixels_against_the_oracle(rt, kernels, shape, knob):
    """NaN / -NaN / +inf / -inf PIXELS through the fused chain, DIRECTLY against the oracle (the reference's dense convolutions:
    every product is formed, so 0 * inf = NaN -- a non-finite value in one channel reaches all three outputs of midget_rgc's
    channel-diagonal kernel; the diagonal forms of the fused kernels reproduce that with a poison term, csrc/silent_rgb2.h),
    for every instantiation: the symmetric forms (0, the default on the reference's kernels), two-group (64), basic (2), dense (1)
    and the one-pixel kernel (16).  Asserted: the same NaN pattern, the same infinities, finite values within the response
    tolerance -- at corners, on edges, next to each other, in 1 x 1 and 2 x 3 levels, under both flat policies."""
    n, h, w, _ = shape
    rng = np.random.default_rng(h * 1000 + w + knob)
    frames = np.stack([noise_frame(250 + i, h, w, 3) for i in range(n)])
    if h > 8:
        frames[0, 3:9, : max(1, w // 3)] = 0.0            # a flat region: 0 * inf under 'ieee'
    bad = [np.nan, -np.nan, np.inf, -np.inf]
    spots = [(0, 0), (h - 1, w - 1), (0, w // 2), (h // 2, 0)] + [(int(rng.integers(0, h)), int(rng.integers(0, w))) for _ in range(6)]
    for k, (y, x) in enumerate(spots):
        frames[0, y, x, k % 3] = bad[k % 4]
    if h > 20 and w > 20:                                 # two different infinities next to each other, and a whole bad pixel
        frames[0, 15, 15, 0], frames[0, 15, 16, 0] = np.inf, -np.inf
        frames[0, 18, 5] = [np.inf, np.nan, -np.inf]
    for policy in ("ieee", "zero"):
        want = so.rgb_line_end_chain(frames, kernels, policy)
        with rt.tuning(TUNE_RGB, knob):
            got = rt.rgb_line_end(frames, kernels, flat_policy=policy)
        for name, ref in (("orient", "orient"), ("line_end", "padded"), ("value", "value")):
            a, b = got[name], want[ref]
            assert np.array_equal(np.isnan(a), np.isnan(b)), "%s %s knob %d: NaN pattern differs (%d vs %d)" % (
                name, policy, knob, np.isnan(a).sum(), np.isnan(b).sum())
            inf = np.isinf(b)
            assert np.array_equal(np.isinf(a), inf) and np.array_equal(a[inf], b[inf]), (name, policy, knob)
            fin = np.isfinite(b)
            if fin.any():
                assert_close(np.where(fin, a, 0), np.where(fin, b, 0), RTOL, scale=max(float(np.abs(b[fin]).max()), 1.0),
                             what="%s %s non-finite pixels, knob %d" % (name, policy, knob), rel_floor=None)


@pytest.mark.parametrize("extents", [[(90, 224), (45, 112), (23, 56), (12, 28)], [(270, 480)], [(7, 4)], [(91, 452), (33, 8), (1, 4)],
                                     [(182, 228), (3, 116)]])
def test_rgb_chain_16_byte_stores_are_bit_identical(rt, kernels, extents):
    """ST4 (csrc/silent_rgb2.h): on levels whose rows start on 16-byte boundaries the chain writes orient / line_end with three
    buffer_store_dwordx4 per pair of rows (row A parked in LDS for a step) instead of four 12-byte stores.  Pure data movement:
    every bit must equal the default 12-byte form (ST4 = RGB knob bit 7) -- odd heights (a last pair with one row), tiles of one row, widths
    that end inside a wave, inside a 4-pixel group never (widths are multiples of 4), several levels, NaN rows from the 'ieee'
    policy, and the extrema instantiation behind silent_rgb_keypoints (maps + keypoints)."""
    import torch
    packed, levels = ragged_pyramid(rt, 77, extents, c=3, n_frames=2)
    levels[0][0, : min(6, extents[0][0]), : extents[0][1] // 2] = 0.0      # 0 * inf under 'ieee'
    packed = rt.PackedPyramid.from_levels(levels)
    for tall in (0, 8):
        with rt.tuning(TUNE_RGB, 128 | tall):
            b = rt.rgb_line_end(packed, kernels)
        with rt.tuning(TUNE_RGB, tall):
            a = rt.rgb_line_end(packed, kernels)
        for name in ("orient", "line_end", "value"):
            np.testing.assert_array_equal(a[name].data.view(np.int32), b[name].data.view(np.int32), err_msg="%s tall %d" % (name, tall))
    # the extrema instantiation (fused keypoints): device buffers, 16-byte aligned by the allocator
    from pysilent_amd.pipeline import LineEndPipeline
    h, w = 136, 240
    frames = torch.from_numpy(np.stack([noise_frame(300 + i, h, w, 3) for i in range(3)])).cuda()
    outs = []
    for knob in (0, 128):
        pipe = LineEndPipeline((h, w), mode="rgb", n_levels=4, batch=3, selection=True, value_map=False, peak_value_map=False)
        with rt.tuning(TUNE_RGB, knob):
            pipe.step(frames)
            torch.cuda.synchronize()
        outs.append(pipe.outputs())
    for name in ("orient", "line_end"):
        assert torch.equal(outs[0][name].data.view(torch.int32), outs[1][name].data.view(torch.int32)), name
    for x, y in zip(outs[0]["keypoints"], outs[1]["keypoints"]):
        np.testing.assert_array_equal(x, y)


def test_rgb_chain_output_subsets(rt, kernels):
    """NULL output pointers: the pair kernel gives an absent map a buffer resource of 0 records (every store of it is dropped
    by the range check); the maps that ARE requested must not change, and asking for nothing is an error."""
    frames = np.stack([noise_frame(70 + i, 45, 230, 3) for i in range(2)])
    full = rt.rgb_line_end(frames, kernels)
    for want in (("orient",), ("line_end",), ("value",), ("orient", "value"), ("line_end", "value")):
        part = rt.rgb_line_end(frames, kernels, want=want)
        assert set(part) == set(want)
        for name in want:
            np.testing.assert_array_equal(part[name], full[name], err_msg=str(want))
    with pytest.raises(Exception):
        rt.rgb_line_end(frames, kernels, want=())


@pytest.mark.parametrize("root", [0.0, 0.5, 1.0])
def test_rgb_chain_regulation_roots(rt, kernels, root):
    """The fused chain's regulator power (exp2(root * log2 m), powf for root = 0 and denormal m) against the oracle for
    other roots than the reference's 0.1, on frames with flat (m = 0) regions under the 'zero' policy."""
    frames = structured_frame(12, 64, 90, 3)[None]
    got = rt.rgb_line_end(frames, kernels, regulation_root=root, flat_policy="zero")
    want = so.rgb_line_end_chain(frames, kernels, flat_policy="zero", blur_root=root)
    bound = eb.rgb_chain(frames, kernels, want, "zero", blur_root=root)
    assert_close(got["orient"], want["orient"], RTOL, what="orient root %g" % root, bound=bound["orient"])
    assert_close(got["line_end"], want["padded"], RTOL, scale=255.0, what="line_end root %g" % root, bound=bound["padded"])


def test_rgb_chain_on_packed_levels_and_keypoints(rt, kernels):
    from pysilent_amd.util.selection import max_value_indices_region
    extents = [(64, 96), (32, 48), (16, 24)]
    packed, levels = ragged_pyramid(rt, 31, extents, c=3, n_frames=2)
    got = rt.rgb_line_end(packed, kernels)
    regions = [(h // 2, w // 2) for h, w in extents]
    kp = max_value_indices_region(got["line_end"], regions, got["value"])
    assert len(kp) == 2
    for f in range(2):
        rows = []
        for l, lev in enumerate(levels):
            v = np.ascontiguousarray(got["value"].level(l)[f:f + 1])
            want = so.rgb_line_end_chain(lev[f:f + 1], kernels)
            assert_close(got["line_end"].level(l)[f:f + 1], want["padded"], RTOL, scale=255.0, what="level %d" % l,
                         bound=eb.rgb_chain(lev[f:f + 1], kernels, want)["padded"])
            r = so.max_value_indices_region(None, (1,) + regions[l] + (3,), v)
            r[:, 0] = l
            rows.append(r)
        np.testing.assert_array_equal(kp[f], np.concatenate(rows))     # bit-exact, row-major, level-major


@pytest.mark.parametrize("c", [1, 3])
@pytest.mark.parametrize("with_value", [True, False])
def test_select_peaks_equals_the_three_separate_ops(rt, c, with_value):
    """silent_select_peaks == top_value_points -> nms3x3(product) -> value_from_color, bit for bit, ragged levels
    (including 1 x 1 and extents around the 60-column wave tile), ties and zero plateaus included."""
    from pysilent_amd.util.selection import top_value_points
    from pysilent_amd.util.color import get_value_from_color
    extents = [(37, 131), (70, 60), (33, 61), (1, 1), (2, 300), (64, 241)]
    rng = np.random.default_rng(5 + c)
    levels = [np.floor(rng.random((2, h, w, c)) * 8).astype(np.float32) * 32 for h, w in extents]      # many ties
    packed = rt.PackedPyramid.from_levels(levels)
    value = get_value_from_color(packed) if with_value else None
    got = rt.select_peaks(packed, 0.1, value)
    top = top_value_points(packed, 0.1, value)
    peaks = rt.nms3x3(top, "product")
    pv = get_value_from_color(peaks)
    np.testing.assert_array_equal(got["top"].data, top.data)
    np.testing.assert_array_equal(got["peaks"].data, peaks.data)
    np.testing.assert_array_equal(got["peak_value"].data, pv.data)
    only = rt.select_peaks(packed, 0.1, value, want=("peak_value",))
    np.testing.assert_array_equal(only["peak_value"].data, pv.data)
    for l, lev in enumerate(levels):                                     # and against the oracle
        v = so.value_from_color(lev)
        np.testing.assert_array_equal(got["peaks"].level(l), so.nms3x3(so.top_value_points(lev, 0.1, v), "product"))


@pytest.mark.parametrize("keep", [False, True])
def test_rgb_pipeline_composite_selection_equals_separate_calls(rt, kernels, keep):
    """silent_select_keypoints (cell maxima folded into the selection pass) gives the same peak value and the same
    keypoints as silent_select_peaks + silent_max_value_indices_region, on levels that cross region cuts inside a tile."""
    import torch
    from pysilent_amd.pipeline import LineEndPipeline
    frames = torch.from_numpy(np.stack([noise_frame(80 + i, 150, 260, 3) for i in range(3)])).cuda()
    pipe = LineEndPipeline((150, 260), mode="rgb", n_levels=4, batch=3, selection=True, keep_selection_maps=keep,
                           max_keypoints_per_frame=1 << 16)
    pipe.step(frames)
    torch.cuda.synchronize()
    out = pipe.outputs()
    for f in range(3):
        rows = []
        for l, (h, w) in enumerate(pipe.extents):
            line = np.ascontiguousarray(out["line_end"].level(l)[f:f + 1].cpu().numpy())
            value = np.ascontiguousarray(out["value"].level(l)[f:f + 1].cpu().numpy())
            pv = so.value_from_color(so.nms3x3(so.top_value_points(line, 0.1, value), "product"))
            np.testing.assert_array_equal(out["peak_value"].level(l)[f:f + 1].cpu().numpy(), pv)
            r = so.max_value_indices_region(None, (1, max(h // 2, 1), max(w // 2, 1), 3), pv)
            r[:, 0] = l
            rows.append(r)
        np.testing.assert_array_equal(out["keypoints"][f], np.concatenate(rows))


@pytest.mark.parametrize("knob", [0, 8, 16, 1])
@pytest.mark.parametrize("value_map", [True, False])
def test_rgb_keypoints_composite_equals_chain_then_selection(rt, kernels, knob, value_map):
    """silent_rgb_keypoints (chain + a-10 -> a-9 -> a-8 -> a-11 in one call; the pair kernel's two-group instantiation
    accumulates the per-level extrema itself and the selection reads the value from line_end) against
    silent_rgb_line_end followed by silent_select_keypoints: every output bit for bit -- with the extrema fused (knobs 0, 8),
    with the one-pixel kernel and with dense weights (16, 1: the composite falls back to the reduction pass), with and
    without the value map, on noise, on line drawings under the 'ieee' policy (NaN regions) and with NaN / inf pixels."""
    import torch
    from pysilent_amd.pipeline import LineEndPipeline
    frames = np.stack([noise_frame(31, 150, 260, 3), structured_frame(32, 150, 260, 3), noise_frame(33, 150, 260, 3)])
    frames[2, 40:60, 100:140] = 0.0
    frames[2, 70, 30, 1] = np.nan
    frames[2, 90, 200, 0] = np.inf
    fused = LineEndPipeline((150, 260), mode="rgb", n_levels=4, batch=3, selection=True, value_map=value_map,
                            max_keypoints_per_frame=1 << 16)
    plain = LineEndPipeline((150, 260), mode="rgb", n_levels=4, batch=3, selection=True, max_keypoints_per_frame=1 << 16)
    t = torch.from_numpy(frames).cuda()
    with rt.tuning(TUNE_RGB, knob):
        fused.step(t)                       # run_pyramid + silent_rgb_keypoints_dev
        plain.run_pyramid(t)
        plain.run_filters()                 # silent_rgb_line_end_dev
        plain.run_keypoints()               # silent_select_keypoints_dev
        torch.cuda.synchronize()
    a, b = fused.outputs(), plain.outputs()
    assert ("value" in a) == value_map
    for name in ("orient", "line_end", "peak_value") + (("value",) if value_map else ()):
        x, y = a[name].data.cpu().numpy(), b[name].data.cpu().numpy()
        assert np.array_equal(np.isnan(x), np.isnan(y)), name
        np.testing.assert_array_equal(np.nan_to_num(x, nan=7.0), np.nan_to_num(y, nan=7.0), err_msg=name)
    np.testing.assert_array_equal(a["keypoint_counts"], b["keypoint_counts"])
    for f in range(3):
        np.testing.assert_array_equal(a["keypoints"][f], b["keypoints"][f])
    assert sum(len(k) for k in a["keypoints"]) > 0


def _sparse_vs_dense(frames, hw, n_levels, flat_policy="ieee", cap=1 << 18, **kw):
    """The fused step with the sparse tail (peak_value_map=False) against the same step with a peak-value map (dense
    kernels): keypoints and counts must be identical; returns (sparse stats, outputs of the dense run)."""
    import torch
    from pysilent_amd.pipeline import LineEndPipeline
    common = dict(mode="rgb", n_levels=n_levels, batch=len(frames), selection=True, value_map=False, flat_policy=flat_policy,
                  max_keypoints_per_frame=cap, **kw)
    sparse = LineEndPipeline(hw, peak_value_map=False, **common)
    mapped = LineEndPipeline(hw, peak_value_map=True, **common)          # sparse machinery + the map (zero fill, scatter)
    dense = LineEndPipeline(hw, peak_value_map=True, **common)
    t = torch.from_numpy(frames).cuda()
    from pysilent_amd import _runtime as rt
    knob = rt.get_context().get_tuning(TUNE_RGB)
    sparse.step(t)
    stats = sparse.sparse_tail_stats()
    mapped.step(t)
    map_stats = mapped.sparse_tail_stats()
    with rt.tuning(TUNE_RGB, knob | 32):                                  # the dense kernels for everything: the reference
        dense.step(t)
        assert not dense.sparse_tail_stats()["ran"]
    torch.cuda.synchronize()
    a, m, b = sparse.outputs(allow_truncated=True), mapped.outputs(allow_truncated=True), dense.outputs(allow_truncated=True)
    assert "peak_value" not in a and "peak_value" in b and map_stats["ran"] == stats["ran"]
    for got in (a, m):
        np.testing.assert_array_equal(got["keypoint_counts"], b["keypoint_counts"])
        for f in range(len(frames)):
            np.testing.assert_array_equal(got["keypoints"][f], b["keypoints"][f])
        for name in ("orient", "line_end") + (("peak_value",) if "peak_value" in got else ()):
            x, y = got[name].data.cpu().numpy(), b[name].data.cpu().numpy()
            assert np.array_equal(np.isnan(x), np.isnan(y)), name
            np.testing.assert_array_equal(np.nan_to_num(x, nan=7.0), np.nan_to_num(y, nan=7.0), err_msg=name)
    return stats, b


@pytest.mark.parametrize("policy", ["ieee", "zero"])
def test_sparse_keypoint_tail_equals_dense_tail(rt, kernels, policy):
    """silent_rgb_keypoints without a peak-value map: selection / NMS / keypoint search evaluated only around the pixels that
    reach their level's threshold.  Same keypoints as the dense kernels on noise, line drawings (NaN regions under 'ieee'),
    NaN / inf pixels, a black frame (every window empty: every pixel a keypoint -> dense fallback), a constant frame and a
    frame of tied plateaus."""
    frames = np.stack([noise_frame(31, 150, 260, 3), structured_frame(32, 150, 260, 3), noise_frame(33, 150, 260, 3),
                       np.zeros((150, 260, 3), np.float32), np.full((150, 260, 3), 90.0, np.float32),
                       structured_frame(34, 150, 260, 3, n_lines=40)])
    frames[2, 40:60, 100:140] = 0.0
    frames[2, 70, 30, 1] = np.nan
    frames[2, 90, 200, 0] = np.inf
    frames[5] = np.round(frames[5] / 64.0) * 64.0          # few distinct colours: ties between line ends
    stats, dense = _sparse_vs_dense(frames, (150, 260), 4, flat_policy=policy)
    assert stats["ran"] and stats["pairs"] == 6 * 4
    # the black frame has windows without a positive peak: all NaN under 'ieee' -> dense kernels; all 0 under 'zero' -> the
    # count pass synthesises the map; the noise frame's large levels are settled by their candidates alone
    assert 0 < stats["dense_pairs"] + stats["zero_map_pairs"] < stats["pairs"]
    assert stats["dense_pairs" if policy == "ieee" else "zero_map_pairs"] >= 4
    assert stats["candidates"] > 0
    counts = dense["keypoint_counts"]
    # the black frame: all NaN under 'ieee' (0 * inf; a NaN is never a keypoint), all 0 under 'zero' (every pixel a keypoint)
    assert counts[0] > 0 and counts[3] == (0 if policy == "ieee" else sum(h * w for h, w in [(150, 260), (75, 130), (38, 65), (19, 32)]))


def test_sparse_keypoint_tail_knob_and_tile_heights(rt, kernels):
    """The summary geometry follows the chain launch's tile height (groups of 16 rows inside a tile): odd tile heights, tiles
    shorter than a group, and the knob that switches the sparse tail off."""
    frames = np.stack([noise_frame(41, 97, 233, 3), structured_frame(42, 97, 233, 3)])
    for knob in (0, (9 << 8), (7 << 8), (20 << 8), (45 << 8), 8):       # tile heights 18 (model), 18, 14, 40, 90, 90
        with rt.tuning(TUNE_RGB, knob):
            stats, _ = _sparse_vs_dense(frames, (97, 233), 3)
        assert stats["ran"], knob
    with rt.tuning(TUNE_RGB, 32):
        stats, _ = _sparse_vs_dense(frames, (97, 233), 3)
    assert not stats["ran"]
    with rt.tuning(TUNE_RGB, 16):                                        # one-pixel kernel: no summary -> dense tail
        stats, _ = _sparse_vs_dense(frames, (97, 233), 3)
    assert not stats["ran"]


def test_sparse_keypoint_tail_candidate_overflow(rt, kernels):
    """More than 16384 pixels reach the threshold in one frame (a regular grid of identical line ends): that frame runs the
    dense kernels, the others stay sparse; keypoints identical."""
    h, w = 400, 640
    grid = np.zeros((h, w, 3), np.float32)
    grid[::4, ::4] = (255.0, 128.0, 64.0)                   # 16 000 isolated dots, each with several line-end responses
    frames = np.stack([grid, noise_frame(43, h, w, 3)])
    stats, dense = _sparse_vs_dense(frames, (h, w), 2, flat_policy="zero", cap=h * w)
    assert stats["ran"] and stats["candidates"] > 16384 and stats["dense_pairs"] >= 2      # both levels of the grid frame


def test_sparse_keypoint_tail_1080p(rt, kernels):
    """Full size (what bench.py times for config 3): 1080p, 6 levels, two frames."""
    frames = np.stack([noise_frame(2, 1080, 1920, 3), structured_frame(3, 1080, 1920, 3)])
    stats, dense = _sparse_vs_dense(frames, (1080, 1920), 6)
    assert stats["ran"] and stats["candidates"] < 20000
    assert dense["keypoint_counts"].min() > 0


def test_rgb_pipeline_with_selection_stage(rt, kernels):
    """chain -> top-percent (a-10, p = 0.1) -> NMS (a-9) -> value -> keypoints (a-11); every stage
    after the chain is index-like and compared bit for bit with the oracle applied to the GPU's own line-end map."""
    import torch
    from pysilent_amd.pipeline import LineEndPipeline
    pipe = LineEndPipeline((96, 160), mode="rgb", n_levels=3, batch=2, selection=True, keep_selection_maps=True,
                           max_keypoints_per_frame=1 << 15)
    frames = np.stack([noise_frame(70 + i, 96, 160, 3) for i in range(2)])      # noise frames stay finite (no 0 * inf)
    pipe.step(torch.from_numpy(frames).cuda())
    torch.cuda.synchronize()
    out = pipe.outputs()
    host = lambda name, l, f: np.ascontiguousarray(out[name].level(l)[f:f + 1].cpu().numpy())
    for f in range(2):
        rows = []
        for l, (h, w) in enumerate(pipe.extents):
            line, value = host("line_end", l, f), host("value", l, f)
            top = so.top_value_points(line, 0.1, value)
            peaks = so.nms3x3(top, "product")
            pv = so.value_from_color(peaks)
            np.testing.assert_array_equal(host("top", l, f), top)
            np.testing.assert_array_equal(host("peaks", l, f), peaks)
            np.testing.assert_array_equal(host("peak_value", l, f), pv)
            r = so.max_value_indices_region(None, (1, max(h // 2, 1), max(w // 2, 1), 3), pv)
            r[:, 0] = l
            rows.append(r)
        np.testing.assert_array_equal(out["keypoints"][f], np.concatenate(rows))


# ----------------------------------------------------------------------------- device-resident (torch) path

def test_torch_device_path_is_bit_identical(rt, kernels):
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available()
    from pysilent_amd.util.zoom import classic_pyramid
    frames = np.stack([noise_frame(s, 135, 240, 1) for s in range(3)])
    host = classic_pyramid(frames, 2.0, 4)
    dev = classic_pyramid(torch.from_numpy(frames).cuda(), 2.0, 4)
    assert dev.on_device
    np.testing.assert_array_equal(dev.data.cpu().numpy(), host.data)
    cs_h, end_h = rt.gray_line_end(host, kernels["cs_gray"], kernels["end4"])
    cs_d, end_d = rt.gray_line_end(dev, kernels["cs_gray"], kernels["end4"])
    np.testing.assert_array_equal(cs_d.data.cpu().numpy(), cs_h.data)
    np.testing.assert_array_equal(end_d.data.cpu().numpy(), end_h.data)
    x = torch.from_numpy(noise_frame(1, 40, 60, 3)[None]).cuda()
    from pysilent_amd import filters
    y = filters.rgc_filter(x)
    assert y.is_cuda
    np.testing.assert_array_equal(y.cpu().numpy(), filters.rgc_filter(x.cpu().numpy()))


# ----------------------------------------------------------------------------- whole gray pass (fused level 0)

@pytest.mark.parametrize("shape,scale,n,K", [((135, 240, 1), 2.0, 5, 4), ((97, 131, 1), 1.7, 4, 8), ((64, 300, 1), 2.0, 3, 3),
                                             ((33, 57, 1), 2.0, 1, 4),
                                             ((270, 480, 1), 2.0, 8, 4),            # 7 general levels: stream kernel <K, 7>
                                             ((270, 480, 1), math.e ** .5, 6, 8),   # the reference's zoom ratio
                                             ((100, 260, 1), 1.2, 3, 4),            # step 1.2: not stream-eligible -> region path
                                             ((270, 480, 1), 2 ** .5, 8, 4),        # round 5: the dense slot layout (stream kernel <K, 7, 1>):
                                             ((135, 240, 1), 1.5, 4, 8),            #   five rows of the first level in flight, ratios down to 1.4
                                             ((216, 384, 1), 2 ** .5, 6, 3)])
def test_gray_pass_equals_pyramid_then_filters(rt, kernels, shape, scale, n, K):
    """silent_gray_pass (level 0 smoothed + filtered in one kernel) is bit-identical to the two-step path,
    and both match the oracle."""
    from pysilent_amd.util.zoom.from_image import classic_levels
    frames = np.stack([noise_frame(s, *shape) for s in range(2)])
    plan = rt.PyramidPlan(shape[0], shape[1], 1, classic_levels(shape[:2], scale, n))
    assert plan.streamable == (n > 1 and scale >= 1.4)      # which path silent_gray_pass takes for this plan
    bank = kernels["end%d" % K]
    pyr, cs, end = plan.gray_pass(frames, kernels["cs_gray"], bank)
    pyr2 = plan.run(frames)
    cs2, end2 = rt.gray_line_end(pyr2, kernels["cs_gray"], bank)
    np.testing.assert_array_equal(pyr.data, pyr2.data)
    np.testing.assert_array_equal(cs.data, cs2.data)
    np.testing.assert_array_equal(end.data, end2.data)
    want = so.classic_pyramid(frames[1], scale, n)
    for l, (wcs, wend) in enumerate(so.gray_line_end_pass(want, kernels["cs_gray"], bank)):
        assert_gray_level_close(pyr.level(l)[1:2], cs.level(l)[1:2], end.level(l)[1:2], want[l], wcs, wend, kernels["cs_gray"], bank, "%d" % l)


def test_gray_pass_reference_layout_and_device_path(rt, kernels):
    """Reference layout (level 0 = centred crop at zoom 1, other levels crops at e^-s/2) through the fused pass,
    host and torch-device paths."""
    torch = pytest.importorskip("torch")
    from pysilent_amd.util.zoom.from_image import reference_levels
    img = noise_frame(5, 240, 320, 1)
    levels = reference_levels((240, 320), (80, 60), math.e ** .5)
    plan = rt.PyramidPlan(240, 320, 1, levels)
    assert not plan.streamable                     # levels are different crops: region + unit-fused + filter kernels
    pyr, cs, end = plan.gray_pass(img[None], kernels["cs_gray"], kernels["end4"])
    want = so.zoom_from_image(img, 1, (80, 60), math.e ** .5)
    got = pyr.data.reshape(want.shape)
    wcs = so.conv2d_same(want, kernels["cs_gray"], relu=True)
    wend = so.conv2d_same(wcs, kernels["end4"], relu=True, clip_hi=255.0)
    assert_gray_level_close(got, cs.data.reshape(wcs.shape), end.data.reshape(wend.shape), want, wcs, wend, kernels["cs_gray"],
                            kernels["end4"], "reference layout")
    dp, dc, de = plan.gray_pass(torch.from_numpy(img[None]).cuda(), kernels["cs_gray"], kernels["end4"])
    np.testing.assert_array_equal(dp.data.cpu().numpy(), pyr.data)
    np.testing.assert_array_equal(dc.data.cpu().numpy(), cs.data)
    np.testing.assert_array_equal(de.data.cpu().numpy(), end.data)


@pytest.mark.parametrize("shape,n", [((1, 1, 1), 1), ((4, 9, 1), 2), ((5, 5, 1), 2), ((6, 70, 1), 3), ((16, 56, 1), 2),
                                     ((17, 57, 1), 3), ((24, 224, 1), 2), ((25, 225, 1), 2)])
def test_gray_pass_tiny_and_tile_boundary_frames(rt, kernels, shape, n):
    """Frames smaller than the stencil reach (levels with < 5 source pixels per axis leave the streaming kernels),
    and extents exactly at / one past the 16 x 56 tile of a wave and the 224-column tile of a block."""
    from pysilent_amd.util.zoom.from_image import classic_levels
    frames = np.stack([noise_frame(40 + s_, *shape) for s_ in range(3)])
    plan = rt.PyramidPlan(shape[0], shape[1], 1, classic_levels(shape[:2], 2.0, n))
    pyr, cs, end = plan.gray_pass(frames, kernels["cs_gray"], kernels["end4"])
    pyr2 = plan.run(frames)
    cs2, end2 = rt.gray_line_end(pyr2, kernels["cs_gray"], kernels["end4"])
    for a, b in ((pyr, pyr2), (cs, cs2), (end, end2)):
        np.testing.assert_array_equal(a.data, b.data)
    want = so.classic_pyramid(frames[2], 2.0, n)
    for l, (wcs, wend) in enumerate(so.gray_line_end_pass(want, kernels["cs_gray"], kernels["end4"])):
        assert_gray_level_close(pyr.level(l)[2:3], cs.level(l)[2:3], end.level(l)[2:3], want[l], wcs, wend, kernels["cs_gray"],
                                kernels["end4"], "tiny %d" % l)


def _nonfinite_frame(seed, h, w, c):
    """Noise frame with NaN / -NaN / +-inf pixels where the unit-level kernels' sixth taps come from somewhere special: the frame's
    corners and edges (mirrored taps: the pixel 3 from the far edge is the last output's sixth tap), the column right of a wave's 64
    lanes and the row below a tile's 24 streamed rows (gray_stream_kernel: 56-column wave tiles, 16-row tiles), neighbours of
    opposite sign (inf - inf), and a 2^-53 leak: a lone huge pixel on a black patch."""
    img = noise_frame(seed, h, w, c)
    neg_nan = np.frombuffer(np.uint32(0xffc00000).tobytes(), np.float32)[0]
    spots = [(0, 0, np.nan), (0, w - 1, np.inf), (h - 1, 0, -np.inf), (h - 1, w - 1, neg_nan), (h - 4, w - 4, np.inf), (3, 3, -np.inf),
             (h // 2, 0, np.nan), (0, w // 2, np.inf), (h // 2, w - 1, -np.inf), (h - 1, w // 2, np.nan)]
    for x in (55, 56, 59, 60, 61, 116, 224 + 60):        # around the edge column of the first waves / the first block
        if x < w - 8:
            spots.append((min(h - 6, 9 + (x % 7)), x, np.inf if x % 2 else np.nan))
    for y in (15, 16, 19, 20, 21, 36):                   # around the extra streamed row of the first tile rows
        if y < h - 8 and w > 40:
            spots.append((y, 30 + (y % 5), -np.inf if y % 2 else np.nan))
    if h > 60 and w > 100:
        spots += [(h // 2 + 7, w // 2, np.inf), (h // 2 + 7, w // 2 + 1, -np.inf)]
        img[h // 2 - 20:h // 2 - 8, 20:36] = 0.0
        spots.append((h // 2 - 14, 30, 3e38))
    for i, (y, x, v) in enumerate(spots):
        img[y, x, i % c] = v
    return img


@pytest.mark.parametrize("shape,scale,n,K", [((96, 160, 1), 2.0, 3, 4), ((135, 300, 1), 2.0, 8, 8), ((135, 300, 1), math.e ** .5, 5, 3),
                                             ((100, 260, 1), 1.2, 3, 8),
                                             ((135, 300, 1), 2 ** .5, 6, 4),   # the dense slot layout <K, 7, 1>
                                             ((109, 216, 1), 2.0, 4, 4)])   # levels whose last row / column is scipy's mode-'constant' artefact
def test_gray_pass_nonfinite_pixels_reach_scipys_six_taps(rt, kernels, shape, scale, n, K):
    """A NaN / inf FRAME pixel poisons what it poisons in the reference: at zoom 1 scipy.ndimage.zoom(order=5) multiplies SIX taps per
    axis (from_image.py:55-59; the sixth weight is 2^-53), so the pixel reaches the level-0 outputs p - 3 .. p + 2 of both axes, and
    everything downstream of them.  Whole maps against the oracle (scipy itself + the filters): same NaN pattern, same infinities,
    finite values inside their bounds -- on the single-read stream kernel <K, 4> / <K, 7>, the unit-fused + region path (zoom step
    1.2) and the two-step path, which must agree with each other bit for bit."""
    from pysilent_amd.util.zoom.from_image import classic_levels
    bad = _nonfinite_frame(3, *shape)
    plan = rt.PyramidPlan(shape[0], shape[1], 1, classic_levels(shape[:2], scale, n))
    assert plan.streamable == (scale > 1.25)
    bank = kernels["end%d" % K]
    got = plan.gray_pass(bad[None], kernels["cs_gray"], bank)
    pyr2 = plan.run(bad[None])
    two = (pyr2,) + tuple(rt.gray_line_end(pyr2, kernels["cs_gray"], bank))
    for a, b in zip(got, two):
        np.testing.assert_array_equal(np.isnan(a.data), np.isnan(b.data))
        np.testing.assert_array_equal(np.nan_to_num(a.data, nan=-1.0), np.nan_to_num(b.data, nan=-1.0))
    with rt.tuning(TUNE_PYRAMID, 1):                     # unit + region kernels instead of pyramid_stream_kernel
        pyr3 = plan.run(bad[None])
    np.testing.assert_array_equal(np.nan_to_num(pyr3.data, nan=-1.0), np.nan_to_num(pyr2.data, nan=-1.0))
    with np.errstate(invalid="ignore", over="ignore"):
        want = so.classic_pyramid(bad, scale, n)
        chain = so.gray_line_end_pass(want, kernels["cs_gray"], bank)
    # the footprint itself, spelled out on the isolated interior NaNs: outputs p - 3 .. p + 2 on both axes, in scipy and here
    g0, w0 = got[0].level(0)[0, :, :, 0], want[0][0, :, :, 0]
    nonfin = ~np.isfinite(bad[:, :, 0])
    seen = 0
    for y, x in zip(*np.nonzero(np.isnan(bad[:, :, 0]))):
        if 8 <= y < shape[0] - 8 and 8 <= x < shape[1] - 8 and nonfin[y - 7:y + 8, x - 7:x + 8].sum() == 1:
            for m in (g0, w0):
                assert np.isnan(m[y - 3:y + 3, x - 3:x + 3]).all() and np.isfinite(m[y - 4:y + 4, x - 4]).all()
                assert np.isfinite(m[y - 4, x - 4:x + 4]).all() and np.isfinite(m[y + 3, x - 4:x + 4]).all() and np.isfinite(m[y - 4:y + 4, x + 3]).all()
            seen += 1
    assert seen >= 2
    for l, (wcs, wend) in enumerate(chain):
        with np.errstate(invalid="ignore", over="ignore"):
            assert_gray_level_close(got[0].level(l)[0:1], got[1].level(l)[0:1], got[2].level(l)[0:1], want[l], wcs, wend,
                                    kernels["cs_gray"], bank, "nonfinite %d" % l)


def test_rgb_pyramids_nonfinite_pixels_reach_scipys_six_taps(rt):
    """The same on three channels: the strip-walk kernel (classic pyramid: 36 pixels per wave; zoom step e^0.5: 32) and the unit +
    region kernels agree bit for bit and show scipy's footprint."""
    from pysilent_amd.util.zoom.from_image import classic_levels
    # (135 x 240 at e^0.5: an outer weight of the last column underflows float32 -- the host keeps it non-zero so that inf * w stays
    # inf; 109 x 220: level 1's last column and level 2 / 3's last rows are scipy's mode-'constant' artefact, exactly 0 whatever the
    # pixels under them hold)
    for shape, scale, n in (((96, 160, 3), 2.0, 4), ((135, 240, 3), math.e ** .5, 5), ((109, 220, 3), 2.0, 4)):
        bad = _nonfinite_frame(5, *shape)
        plan = rt.PyramidPlan(shape[0], shape[1], 3, classic_levels(shape[:2], scale, n))
        assert plan.walk_plans[0] == 1
        pyr = plan.run(bad[None])
        with rt.tuning(TUNE_PYRAMID, 2):
            pyr2 = plan.run(bad[None])
        np.testing.assert_array_equal(np.nan_to_num(pyr.data, nan=-1.0), np.nan_to_num(pyr2.data, nan=-1.0))
        with np.errstate(invalid="ignore", over="ignore"):
            want = so.classic_pyramid(bad, scale, n)
            for l in range(n):
                assert_close(pyr.level(l)[0:1], want[l], RTOL, scale=255.0, what="rgb nonfinite pyramid %d" % l, bound=eb.zoom(want[l]))


def test_gray_filters_nan_and_inf_propagate_like_the_oracle(rt, kernels):
    """Non-finite values injected into a finite pyramid: a NaN stays a NaN through relu / clip (Eigen's (x < 0) ? 0 : x), +inf clips
    to 255 -- every level, against the oracle."""
    from pysilent_amd.util.zoom.from_image import classic_levels
    frame = noise_frame(3, 96, 160, 1)
    plan = rt.PyramidPlan(96, 160, 1, classic_levels((96, 160), 2.0, 3))
    # the filters themselves: non-finite values injected into a finite pyramid, every level, against the oracle
    pyr = plan.run(frame[None])
    levels = [np.array(pyr.level(l)) for l in range(3)]
    levels[0][0, 30, 40, 0] = np.nan
    levels[1][0, 5, 7, 0] = np.inf
    levels[2][0, 20, 3, 0] = -np.inf
    packed = rt.PackedPyramid.from_levels(levels)
    cs, end = rt.gray_line_end(packed, kernels["cs_gray"], kernels["end4"])
    for l, lev in enumerate(levels):
        wcs = so.conv2d_same(lev, kernels["cs_gray"], relu=True)
        wend = so.conv2d_same(wcs, kernels["end4"], relu=True, clip_hi=255.0)
        e_cs, e_end = eb.gray_chain(lev, kernels["cs_gray"], kernels["end4"], wcs)
        assert_close(cs.level(l), wcs, RTOL, scale=255.0, what="cs %d" % l, bound=e_cs)          # NaN pattern checked inside
        assert_close(end.level(l), wend, RTOL, scale=255.0, what="end %d" % l, bound=e_end)
    assert np.isnan(end.level(0)).any() and np.nanmax(end.level(1)) <= 255.0


def test_config5_4k_8_levels_8_orientations_against_c_oracle(rt, kernels):
    """BASELINE config 5 (the largest): one 3840 x 2160 frame, 8-level pyramid, K = 8 -- stream kernel <8, 7>."""
    from pysilent_amd.util.zoom.from_image import classic_levels
    frame = structured_frame(9, 2160, 3840, 1, n_lines=2000)
    plan = rt.PyramidPlan(2160, 3840, 1, classic_levels((2160, 3840), 2.0, 8))
    assert plan.streamable and len(plan.extents) == 8
    pyr, cs, end = plan.gray_pass(frame[None], kernels["cs_gray"], kernels["end8"])
    want_pyr = co.classic_pyramid(frame, plan.extents)
    for l in range(8):
        wcs, wend = co.gray_line_end_level(want_pyr[l], kernels["cs_gray"], kernels["end8"])
        assert_gray_level_close(pyr.level(l), cs.level(l), end.level(l), want_pyr[l], wcs, wend, kernels["cs_gray"], kernels["end8"], "config5 %d" % l)


def test_config3_1080p_rgb_full_size_against_c_oracle(rt, kernels):
    """BASELINE config 3 at full size, one frame: 1080p RGB, 6-level pyramid, rgc > rgby > stripe > regulate > end > pad >
    value, top 10 %, NMS, keypoints.  Chain maps within tolerance of the C oracle; every index-like stage bit-exact
    against the oracle applied to the GPU's own line-end map."""
    import torch
    from pysilent_amd.pipeline import LineEndPipeline
    frame = noise_frame(2, 1080, 1920, 3)
    pipe = LineEndPipeline((1080, 1920), mode="rgb", n_levels=6, batch=1, selection=True, keep_selection_maps=True,
                           max_keypoints_per_frame=1 << 18)
    pipe.step(torch.from_numpy(frame[None]).cuda())
    torch.cuda.synchronize()
    out = pipe.outputs()
    host = lambda name, l: np.ascontiguousarray(out[name].level(l).cpu().numpy())
    want_pyr = co.classic_pyramid(frame, pipe.extents)
    rows = []
    for l, (h, w) in enumerate(pipe.extents):
        assert_close(host("pyramid", l), want_pyr[l], RTOL, scale=255.0, what="pyramid %d" % l, bound=eb.zoom(want_pyr[l]))
        chain = {}
        x = want_pyr[l]
        for name in ("rgc", "rgby", "stripe"):
            x = chain[name] = co.conv2d_same(x, kernels[name], relu=True)
        orient = chain["orient"] = co.regulate(x, kernels["blur"], 1.0, 0.1)
        line = chain["padded"] = co.pad_inwards(co.conv2d_same(orient, kernels["end"], relu=True, clip_hi=255.0), [[0, 0], [2, 2], [2, 2], [0, 0]])
        bound = eb.rgb_chain(want_pyr[l], kernels, chain, e_x=eb.zoom(want_pyr[l]))
        assert_close(host("orient", l), orient, RTOL, what="orient %d" % l, bound=bound["orient"])
        assert_close(host("line_end", l), line, RTOL, scale=255.0, what="line_end %d" % l, bound=bound["padded"])
        g_line, g_val = host("line_end", l), host("value", l)
        np.testing.assert_array_equal(g_val, co.value_from_color(g_line))
        top = co.top_value_points(g_line, 0.1, g_val)
        peaks = co.nms3x3(top, "product")
        pv = co.value_from_color(peaks)
        np.testing.assert_array_equal(host("top", l), top)
        np.testing.assert_array_equal(host("peaks", l), peaks)
        np.testing.assert_array_equal(host("peak_value", l), pv)
        r = co.max_value_indices_region(None, (1, max(h // 2, 1), max(w // 2, 1), 3), pv)
        r[:, 0] = l
        rows.append(r)
    np.testing.assert_array_equal(out["keypoints"][0], np.concatenate(rows))


# ----------------------------------------------------------------------------- full size (BASELINE config 2)

def test_config2_1080p_full_size_against_c_oracle(rt, kernels):
    """1080p gray, 5-level pyramid, CS + 4-orientation line-end: the bench workload, one frame."""
    from pysilent_amd.util.zoom import classic_pyramid
    frame = noise_frame(0, 1080, 1920, 1)
    pyr = classic_pyramid(frame, 2.0, 5)
    assert pyr.extents == [(1080, 1920), (540, 960), (270, 480), (135, 240), (68, 120)]
    assert pyr.frame_px == 2762160
    cs, end = rt.gray_line_end(pyr, kernels["cs_gray"], kernels["end4"])
    from pysilent_amd.util.zoom.from_image import classic_levels
    fused = rt.PyramidPlan(1080, 1920, 1, classic_levels((1080, 1920), 2.0, 5)).gray_pass(
        frame[None], kernels["cs_gray"], kernels["end4"])
    for a, b in zip(fused, (pyr, cs, end)):
        np.testing.assert_array_equal(a.data, b.data)          # the bench path (silent_gray_pass) == two-step path
    want_pyr = co.classic_pyramid(frame, pyr.extents)
    for l in range(5):
        wcs, wend = co.gray_line_end_level(want_pyr[l], kernels["cs_gray"], kernels["end4"])
        assert_gray_level_close(pyr.level(l), cs.level(l), end.level(l), want_pyr[l], wcs, wend, kernels["cs_gray"], kernels["end4"], "config2 %d" % l)
    # size-independent properties: linearity of the pyramid, and ReLU/clip range of the responses
    pyr2 = classic_pyramid(frame * np.float32(0.5), 2.0, 5)
    np.testing.assert_allclose(pyr2.data, pyr.data * np.float32(0.5), rtol=1e-6, atol=1e-4)
    assert end.data.min() >= 0 and end.data.max() <= 255 and cs.data.min() >= 0


# ----------------------------------------------------------------------------- config 4: the per-rank share

def test_config4_per_rank_share_64_frames_against_c_oracle(rt, kernels):
    """BASELINE config 4 = 512 synthetic 1080p frames sharded over 8 GPUs: each rank runs 64-frame batches of the
    config-2 pass through LineEndPipeline.step (what bench.py times).  Rank 3 of 8's share: frames 3, 11, 19, ...;
    frames spread over the batch (first, middle, last, and two more) are checked against the C oracle, and the
    frames in between against a checksum of checksums (frame j of this batch == the same frame run alone)."""
    import torch
    from pysilent_amd import distributed as D
    from pysilent_amd.pipeline import LineEndPipeline
    B, world, rank = 64, 8, 3
    mine = D.shard_frame_indices(B * world, rank, world)
    assert len(mine) == B and mine[:3] == [3, 11, 19]
    pipe = LineEndPipeline((1080, 1920), mode="gray", n_levels=5, n_orient=4, batch=B)
    frames = torch.empty((B, 1080, 1920, 1), dtype=torch.float32, device=pipe.tdev)
    host_frames = {}
    for j, gi in enumerate(mine):
        f = D.synthetic_frame(gi, 1080, 1920, 1)
        if j in (0, 17, 31, 46, 63):
            host_frames[j] = f
        frames[j] = torch.from_numpy(f).to(pipe.tdev)
    pipe.step(frames)
    torch.cuda.synchronize()
    out = pipe.outputs()
    for j, frame in host_frames.items():
        want_pyr = co.classic_pyramid(frame, pipe.extents)
        for l in range(5):
            wcs, wend = co.gray_line_end_level(want_pyr[l], kernels["cs_gray"], kernels["end4"])
            host = lambda name: out[name].level(l)[j:j + 1].cpu().numpy()
            assert_gray_level_close(host("pyramid"), host("cs"), host("end"), want_pyr[l], wcs, wend, kernels["cs_gray"],
                                    kernels["end4"], "config4 %d" % l)
    # every frame of the batch equals that frame run alone (batch position must not matter): compare per-frame sums
    solo = LineEndPipeline((1080, 1920), mode="gray", n_levels=5, n_orient=4, batch=1)
    for j in (1, 2, 30, 62):
        solo.step(frames[j:j + 1])
        torch.cuda.synchronize()
        alone = solo.outputs()
        for name in ("pyramid", "cs", "end"):
            assert torch.equal(out[name].data.view(B, -1)[j], alone[name].data.view(-1)), (name, j)


# ----------------------------------------------------------------------------- NaN through the peak stage

def _nan_bearing_levels(c):
    """Ragged levels with NaN single pixels, NaN rows, a whole NaN quadrant, an all-NaN level and a NaN in every lane
    position of a wave (columns 0 .. 70 of one row)."""
    rng = np.random.default_rng(77 + c)
    extents = [(37, 131), (70, 60), (5, 7), (64, 241)]
    levels = [np.floor(rng.random((2, h, w, c)) * 16).astype(np.float32) * 16 for h, w in extents]
    levels[0][0, 3, 5] = np.nan
    levels[0][1, 10, :] = np.nan
    levels[0][0, 20, np.arange(0, 71, 2)] = np.nan
    levels[1][0, :35, :30] = np.nan                      # a keypoint window of the (35, 30) regions holds only NaN
    levels[2][1] = np.nan                                # a level without a single value
    levels[3][0, 0, 0] = np.nan
    levels[3][1, 63, 240] = np.nan
    levels[3][0, 17, 59:63] = np.nan                     # across the 60-column wave boundary
    return extents, levels


@pytest.mark.parametrize("c", [1, 3])
def test_peak_stage_ignores_nan_like_tf1_max_pool(rt, c):
    """a-10 / a-9 / a-11 on NaN-bearing maps: every max_pool ignores NaN taps (oracle.pool_max cites the TF 1.x kernel),
    a NaN value is never selected and never a keypoint, NaN colours stay NaN.  Bit-exact against the oracle, NaN
    patterns included, for the separate ops, the fused selection pass and the keypoint composite."""
    from pysilent_amd.util.selection import top_value_points, max_value_indices_region
    from pysilent_amd.util.color import get_value_from_color
    extents, levels = _nan_bearing_levels(c)
    packed = rt.PackedPyramid.from_levels(levels)
    value = get_value_from_color(packed)
    top = top_value_points(packed, 0.1, value)
    peaks = rt.nms3x3(top, "product")
    fired = rt.nms3x3(value, "fired")
    fused = rt.select_peaks(packed, 0.1, value)
    regions = [(max(h // 2, 1), max(w // 2, 1)) for h, w in extents]
    for l, lev in enumerate(levels):
        v = so.value_from_color(lev)
        np.testing.assert_array_equal(value.level(l), v)
        wtop = so.top_value_points(lev, 0.1, v)
        wpeaks = so.nms3x3(wtop, "product")
        np.testing.assert_array_equal(top.level(l), wtop)
        np.testing.assert_array_equal(peaks.level(l), wpeaks)
        np.testing.assert_array_equal(fired.level(l), so.nms3x3(v, "fired"))
        np.testing.assert_array_equal(fused["top"].level(l), wtop)
        np.testing.assert_array_equal(fused["peaks"].level(l), wpeaks)
        np.testing.assert_array_equal(fused["peak_value"].level(l), so.value_from_color(wpeaks))
    assert np.isnan(top.level(0)[0, 3, 5]).all() and not np.isnan(top.level(0)[0, 3, 6]).any()
    # keypoints straight from the NaN-bearing value map, and from the NaN-bearing peak value
    for vmap in (value, fused["peak_value"]):
        kp = max_value_indices_region(packed, regions, vmap)
        for f in range(2):
            rows = []
            for l in range(len(levels)):
                r = so.max_value_indices_region(None, (1,) + regions[l] + (c,), np.ascontiguousarray(vmap.level(l)[f:f + 1]))
                r[:, 0] = l
                rows.append(r)
            want = np.concatenate(rows)
            np.testing.assert_array_equal(kp[f], want)
            assert len(want) > 0
            if f == 1:
                assert not (want[:, 0] == 2).any()          # frame 1's level 2 is all NaN: no keypoint there


@pytest.mark.parametrize("keep", [False, True])
def test_config3_chain_on_line_drawings_under_ieee_through_keypoints(rt, kernels, keep):
    """The default flat policy (0 * inf = NaN on every flat region, gaussian_regulator_tensor.py:35-36) on line drawings --
    the frames the detector is for: selection + keypoints after the chain, fused composite (keep=False) and separate
    calls (keep=True) give the same keypoints, bit-exact against the oracle applied to the GPU's own line-end map."""
    import torch
    from pysilent_amd.pipeline import LineEndPipeline
    frames = np.stack([structured_frame(40 + i, 150, 260, 3) for i in range(2)])
    pipe = LineEndPipeline((150, 260), mode="rgb", n_levels=4, batch=2, selection=True, keep_selection_maps=keep,
                           flat_policy="ieee", max_keypoints_per_frame=1 << 16)
    pipe.step(torch.from_numpy(frames).cuda())
    torch.cuda.synchronize()
    out = pipe.outputs()
    n_nan = 0
    for f in range(2):
        rows = []
        for l, (h, w) in enumerate(pipe.extents):
            line = np.ascontiguousarray(out["line_end"].level(l)[f:f + 1].cpu().numpy())
            value = np.ascontiguousarray(out["value"].level(l)[f:f + 1].cpu().numpy())
            n_nan += int(np.isnan(value).sum())
            pv = so.value_from_color(so.nms3x3(so.top_value_points(line, 0.1, value), "product"))
            np.testing.assert_array_equal(out["peak_value"].level(l)[f:f + 1].cpu().numpy(), pv)
            r = so.max_value_indices_region(None, (1, max(h // 2, 1), max(w // 2, 1), 3), pv)
            r[:, 0] = l
            rows.append(r)
        want = np.concatenate(rows)
        np.testing.assert_array_equal(out["keypoints"][f], want)
        assert len(want) > 0                                  # NaN regions no longer silence the level's threshold
    assert n_nan > 1000                                       # the frames really exercise the NaN path


def test_generate_recovery_and_gather_to_host(rt):
    import torch
    from pysilent_amd.util.energy import generate_recovery
    x = np.array([[1.0, 20.0], [300.0, 0.0]], np.float32)[None, :, :, None]
    np.testing.assert_array_equal(generate_recovery(x, True, False), so.generate_recovery(x, True, False))
    np.testing.assert_array_equal(generate_recovery(x, True, True), so.generate_recovery(x, True, True))
    np.testing.assert_array_equal(generate_recovery(x, False, True), so.generate_recovery(x, False, True))
    a = torch.arange(7, dtype=torch.float32, device="cuda")
    b = torch.full((3, 5), 2.5, dtype=torch.float32, device="cuda")
    host = rt.gather_to_host([a, b, a[:0]])
    np.testing.assert_array_equal(host, np.concatenate([np.arange(7, dtype=np.float32), np.full(15, 2.5, np.float32)]))
    with pytest.raises(ValueError):
        rt.gather_to_host([a.double()])


def test_entry_points_leave_the_callers_device_alone(rt):
    """A context call must not move the caller's current HIP device (torch tracks it through hipGetDevice)."""
    import torch
    before = torch.cuda.current_device()
    rt.nms3x3(noise_frame(1, 9, 9, 1)[None], "fired")
    assert torch.cuda.current_device() == before


# ----------------------------------------------------------------------------- a-11 with any region_shape

@pytest.mark.parametrize("shape,region", [((2, 192, 288, 3), (1, 3, 3, 3)),        # the reference's centroid_region_shape as regions: 64 x 96 windows
                                          ((1, 37, 53, 1), (1, 5, 7, 1)),
                                          ((2, 40, 40, 3), (1, 1, 1, 3)),          # every pixel its own stride: 40 x 40 windows
                                          ((1, 64, 300, 1), (1, 64, 7, 1)),        # many windows on one axis only
                                          ((1, 135, 240, 1), (1, 9, 16, 1)),
                                          ((3, 19, 27, 1), (1, 2, 27, 1)),
                                          ((1, 3, 20011, 1), (1, 2, 1501, 1)),     # a level wider than the row pass's 8192-pixel chunk
                                          ((1, 2, 16400, 1), (1, 1, 3000, 1))])    # (round 3 refused levels wider than 16 384 px)
def test_max_value_indices_region_with_many_windows(rt, shape, region):
    """More than 4 windows per axis (the kernarg cell tables of the fast path do not apply): separable prefix / suffix
    window maxima.  Bit-exact against the oracle, ties, zero plateaus and NaNs included, for NHWC tensors and packed levels."""
    from pysilent_amd.util.selection import max_value_indices_region
    x = np.floor(np.random.default_rng(12).random(shape) * 16).astype(np.float32) * 16      # many ties
    x[0, :shape[1] // 3, :shape[2] // 3] = 0
    x[0, shape[1] // 2, shape[2] // 2:] = np.nan
    got = max_value_indices_region(x, region)
    want = so.max_value_indices_region(x, region)
    assert got.dtype == np.int64 and len(want) > 0
    np.testing.assert_array_equal(got, want)
    v = so.value_from_color(x)
    np.testing.assert_array_equal(max_value_indices_region(x, region, v), co.max_value_indices_region(x, region, v))


def test_select_keypoints_with_many_windows(rt):
    """The config-3 composite (top-percent > NMS > value > keypoints) with 3 x 3 regions: the selection pass without folded
    cell maxima + the general window-maxima path, equal to the separate calls and to the oracle."""
    import ctypes as C
    from pysilent_amd import _lib
    from pysilent_amd.util.selection import max_value_indices_region
    extents = [(37, 131), (70, 60), (5, 7)]
    rng = np.random.default_rng(3)
    levels = [np.floor(rng.random((2, h, w, 3)) * 8).astype(np.float32) * 32 for h, w in extents]
    packed = rt.PackedPyramid.from_levels(levels)
    fused = rt.select_peaks(packed, 0.1)
    regions = [(3, 3)] * 3
    kp = max_value_indices_region(packed, regions, fused["peak_value"])
    for f in range(2):
        rows = []
        for l, lev in enumerate(levels):
            pv = so.value_from_color(so.nms3x3(so.top_value_points(lev[f:f + 1], 0.1), "product"))
            r = so.max_value_indices_region(None, (1, 3, 3, 3), pv)
            r[:, 0] = l
            rows.append(r)
        np.testing.assert_array_equal(kp[f], np.concatenate(rows))
    # the composite entry point through the C ABI (host twin)
    ctx = rt.get_context()
    lib = _lib.load()
    n_levels, n_frames = 3, 2
    lv = (_lib.Extent * n_levels)(*[_lib.Extent(h, w) for h, w in extents])
    rg = (_lib.Extent * n_levels)(*[_lib.Extent(3, 3)] * n_levels)
    cap = packed.frame_px
    pv_out = np.empty(n_frames * packed.frame_px, np.float32)
    idx = np.empty((n_frames, cap, 4), np.int64)
    counts = np.zeros(n_frames, np.int64)
    f32p = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    ctx.check(lib.silent_select_keypoints(ctx.handle, f32p(packed.data), None, lv, n_levels, n_frames, 3, C.c_double(0.1), rg,
                                          f32p(pv_out), C.c_void_p(idx.ctypes.data), C.c_size_t(cap),
                                          C.c_void_p(counts.ctypes.data)))
    np.testing.assert_array_equal(pv_out, fused["peak_value"].data)
    for f in range(2):
        np.testing.assert_array_equal(idx[f, :counts[f]], kp[f])


def test_rgb_keypoints_host_entry_point_with_and_without_the_peak_value_map(rt, kernels):
    """silent_rgb_keypoints, the HOST-pointer twin (pyramid in host memory, maps and int64 rows back): with a peak-value map
    (dense tail) and with peak_value_out = NULL (sparse tail) the same rows, equal to the device pipeline's; and
    silent_select_keypoints with peak_value_out = NULL (the map then lives in the context workspace)."""
    import ctypes as C
    import torch
    from pysilent_amd import _lib
    from pysilent_amd.pipeline import LineEndPipeline
    h, w, n_levels, B = 150, 260, 3, 2
    frames = np.stack([noise_frame(51, h, w, 3), structured_frame(52, h, w, 3)])
    pipe = LineEndPipeline((h, w), mode="rgb", n_levels=n_levels, batch=B, selection=True, value_map=False, peak_value_map=False,
                           max_keypoints_per_frame=h * w * 2)
    pipe.step(torch.from_numpy(frames).cuda())
    torch.cuda.synchronize()
    dev = pipe.outputs()
    pyr = np.ascontiguousarray(dev["pyramid"].data.cpu().numpy())
    ctx, lib = rt.get_context(), _lib.load()
    lv = (_lib.Extent * n_levels)(*[_lib.Extent(eh, ew) for eh, ew in pipe.extents])
    rg = (_lib.Extent * n_levels)(*[_lib.Extent(max(eh // 2, 1), max(ew // 2, 1)) for eh, ew in pipe.extents])
    fp = C.POINTER(C.c_float)
    ks = {k: np.ascontiguousarray(v, np.float32) for k, v in pipe.consts.items()}
    params = _lib.RgbChainParams(*[ks[k].ctypes.data_as(fp) for k in ("rgc", "rgby", "stripe", "blur", "end")], 1.0, 0.1, _lib.FLAT_IEEE, 255.0, 2)
    cap = pipe.frame_px
    got = {}
    for with_map in (True, False):
        line = np.empty(B * pipe.frame_px * 3, np.float32)
        pv = np.empty(B * pipe.frame_px, np.float32)
        idx = np.empty((B, cap, 4), np.int64)
        counts = np.zeros(B, np.int64)
        ctx.check(lib.silent_rgb_keypoints(ctx.handle, pyr.ctypes.data, lv, n_levels, B, C.byref(params), C.c_double(0.1), rg, None,
                                           line.ctypes.data, None, pv.ctypes.data if with_map else None, idx.ctypes.data, cap,
                                           counts.ctypes.data))
        got[with_map] = (line, [idx[f, :counts[f]].copy() for f in range(B)], pv)
    for f in range(B):
        np.testing.assert_array_equal(got[True][1][f], dev["keypoints"][f])
        np.testing.assert_array_equal(got[False][1][f], dev["keypoints"][f])
    a, b = got[True][0], dev["line_end"].data.cpu().numpy()
    np.testing.assert_array_equal(np.nan_to_num(a, nan=7.0), np.nan_to_num(b, nan=7.0))
    # silent_select_keypoints on the host's line_end, without a peak-value map
    idx = np.empty((B, cap, 4), np.int64)
    counts = np.zeros(B, np.int64)
    ctx.check(lib.silent_select_keypoints(ctx.handle, got[True][0].ctypes.data, None, lv, n_levels, B, 3, C.c_double(0.1), rg, None,
                                          idx.ctypes.data, cap, counts.ctypes.data))
    for f in range(B):
        np.testing.assert_array_equal(idx[f, :counts[f]], dev["keypoints"][f])


def test_rgb_chain_on_plateau_frames_under_ieee_by_the_three_zone_rule(rt, kernels):
    """Plateau frames (flat coloured regions whose stencil taps cancel to rounding residue, black regions whose taps are
    exactly 0, and edges between them) under the reference's default flat policy: the regulator's NaN pattern must equal
    the oracle's wherever it is decidable (exactly-zero windows, ordinary responses) and be {NaN, residue} in the band where
    the summation order decides -- the deterministic rule of conftest.assert_regulated_close (this replaces the ad-hoc
    "residue criterion" of round 1's fuzz script).  The stages after the regulator are compared against the oracle applied
    to the GPU's own orient map."""
    rng = np.random.default_rng(21)
    frames = np.floor(rng.random((2, 96, 150, 3)) * 4).astype(np.float32) * 64          # 4-level plateaus, random per pixel
    frames = np.repeat(np.repeat(frames[:, ::6, ::6], 6, axis=1), 6, axis=2)[:, :96, :150]   # 6 x 6 flat blocks
    frames[0, 20:60, 30:90] = 0.0                                                        # a black region
    frames[1, :, 100:] = 128.0                                                           # a large flat grey region
    ks = {x: kernels[x] for x in ("rgc", "rgby", "stripe", "blur", "end")}
    got = rt.rgb_line_end(frames, ks, flat_policy="ieee")
    want = so.rgb_line_end_chain(frames, ks, "ieee")
    b = so.conv2d_same(want["stripe"], ks["blur"])
    n_resid, n_flip = assert_regulated_close(got["orient"], want["stripe"], b, want["orient"], RTOL, what="plateau orient")
    assert np.isnan(want["orient"]).sum() > 1000                 # the black region really produces the 0 * inf case
    le = so.pad_inwards(so.conv2d_same(np.ascontiguousarray(got["orient"]), ks["end"], relu=True, clip_hi=255.0),
                        [[0, 0], [2, 2], [2, 2], [0, 0]])
    assert_close(got["line_end"], le, RTOL, scale=255.0, what="plateau line_end from the GPU's orient",
                 bound=eb.pad(eb.conv(np.ascontiguousarray(got["orient"]), ks["end"]), 2))
    np.testing.assert_array_equal(got["value"], so.value_from_color(np.ascontiguousarray(got["line_end"])))
    print("plateau frames: %d residue-band values, NaN-or-residue differs from the oracle at %d of them" % (n_resid, n_flip))


def test_rccl_calls_execute_on_one_gpu(tmp_path):
    """The RCCL branches of pysilent_amd.distributed (broadcast of the constants, MAX all-reduce, barrier with device_ids)
    in a 1-rank "nccl" process group: RCCL refuses two ranks on one device, so this is as much of the real backend as a
    one-GPU box can run; the 2-rank logic is covered by the gloo tests on CPU."""
    import json, subprocess, sys, textwrap
    from conftest import ROOT
    script = tmp_path / "one_rank.py"
    script.write_text(textwrap.dedent("""
        import json, sys
        sys.path.insert(0, %r)
        import numpy as np, torch, torch.distributed as dist
        from pysilent_amd import distributed as D
        from pysilent_amd.pipeline import default_constants
        rank, world, local = D.init(backend="nccl")
        assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
        c = D.broadcast_constants("gray", 4, device=local)
        ok = all(np.array_equal(c[k], default_constants("gray", 4)[k]) for k in c)
        r = D.broadcast_constants("rgb", device=local)
        D.barrier()
        slow = D.max_over_ranks(2.5)
        print(json.dumps({"ok": bool(ok), "slow": slow, "n_rgb": len(r)}))
        D.finalize()
    """) % ROOT)
    import os, socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               SILENT_DIST_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out == {"ok": True, "slow": 2.5, "n_rgb": 5}


def test_pipeline_flags_keypoint_overflow(rt, kernels):
    """The device entry points are asynchronous and cannot return SILENT_E_CAPACITY; counts > cap is the flag, and
    LineEndPipeline.outputs() turns it into the same ValueError as the host path."""
    import torch
    from pysilent_amd.pipeline import LineEndPipeline
    pipe = LineEndPipeline((64, 96), mode="rgb", n_levels=2, batch=1, max_keypoints_per_frame=5, flat_policy="zero")
    pipe.step(torch.zeros((1, 64, 96, 3), device="cuda"))          # an all-zero frame (zero policy): every pixel is a keypoint
    torch.cuda.synchronize()
    with pytest.raises(ValueError, match="keypoint capacity exceeded"):
        pipe.outputs()
    out = pipe.outputs(allow_truncated=True)
    assert len(out["keypoints"][0]) == 5 and int(out["keypoint_counts"][0]) == 64 * 96 + 32 * 48


# ----------------------------------------------------------------------------- single-read RGB pyramid (silent_walk_rgb.h)

@pytest.mark.parametrize("shape,scale,n,B", [((135, 240, 3), 2.0, 4, 2),        # one strip + a ragged one (240 = 144 + 96)
                                             ((64, 64, 3), 2.0, 3, 3),
                                             ((270, 480, 3), 2.0, 8, 1),         # 7 general levels
                                             ((97, 1000, 3), 2.0, 4, 2),         # 7 strips, the last holds 136 pixels
                                             ((200, 148, 3), 2.5, 3, 2),         # second strip holds 4 pixels
                                             ((40, 8, 3), 2.0, 2, 1),            # narrower than a wave tile
                                             ((135, 241, 3), 2.0, 4, 2),         # widths that are not a multiple of 4 (round 5): rows start on
                                             ((64, 66, 3), 2.0, 3, 3),           # 4, 8 or 12 bytes -- the loader fetches 12-byte pixels
                                             ((97, 999, 3), 2.0, 4, 2),
                                             ((270, 1366, 3), math.e ** .5, 5, 1),
                                             ((1080, 1920, 3), 2.0, 6, 1)])      # config 3 geometry
def test_rgb_pyramid_walk_is_bit_identical_to_unit_plus_region_kernels(rt, shape, scale, n, B):
    """pyramid_walk3_kernel (one read of the frame: loader wave + ring, two floats per lane, other levels gathered from a
    wave-private line) against pyramid_unit_kernel<3> + pyramid_region_kernel<3> (PYRAMID knob 2): same taps, same order,
    every level equal bit for bit -- across strips, segments, ragged edges, NaN / inf pixels -- and both match the oracle."""
    from pysilent_amd.util.zoom.from_image import classic_levels
    frames = np.stack([noise_frame(70 + s_, *shape) for s_ in range(B)])
    frames[0, shape[0] // 2, shape[1] // 3, 1] = np.nan
    frames[B - 1, 0, 0, 2] = np.inf
    plan = rt.PyramidPlan(shape[0], shape[1], 3, classic_levels(shape[:2], scale, n))
    assert plan.walk_plans == ((1, 36 if scale >= 1.875 else 32) if n > 1 else (0, 0))
    got = plan.run(frames)
    with rt.tuning(TUNE_PYRAMID, 2):
        two = plan.run(frames)
    np.testing.assert_array_equal(got.data, two.data)
    clean = noise_frame(99, *shape)
    want = so.classic_pyramid(clean, scale, n)
    res = plan.run(clean[None])
    for l in range(n):
        assert_close(res.level(l), want[l], RTOL, scale=255.0, what="rgb walk level %d" % l, bound=eb.zoom(want[l]))


@pytest.mark.parametrize("shape,center,scale,B", [((1080, 1920, 3), (288, 192), math.e ** .5, 2),    # the reference's defaults on 1080p: 4 plans
                                                  ((480, 640, 3), (288, 192), math.e ** .5, 3),      # ... on its 640 x 480 camera frame: 2 plans
                                                  ((480, 640, 3), (160, 120), math.e ** .5, 1),      # 3 levels
                                                  ((270, 480, 3), (61, 45), 2.0, 2),                 # odd extents and crop offsets (alignment shift 1..3)
                                                  ((300, 500, 3), (100, 37), 1.7, 2),                # one axis clips first
                                                  ((270, 482, 3), (61, 45), 2.0, 2),                 # frame widths that are not a multiple of 4
                                                  ((481, 641, 3), (288, 192), math.e ** .5, 1),
                                                  ((97, 132, 3), (32, 24), 1.7, 1)])
def test_rgb_pyramid_walk_on_the_references_crop_layout(rt, shape, center, scale, B):
    """The reference's own pyramid layout (image_to_zoom_tensor, from_image.py:45-64: nested centre crops resampled to one
    fixed size) through ONE launch of the strip-walk kernel.  Round 5: a plan for the unit level and ONE "union" plan for the other
    levels -- the walk over the outermost crop serves every inner level wherever its taps stay inside that level's crop, the inner
    levels' first / last output rows and columns (scipy mirrors them at the level's OWN crop edge) come from pyramid_border_kernel;
    PYRAMID knob 4 at plan creation: round 3's one plan per level.  Every level equal bit for bit between the union plans, the
    per-level plans and the unit + region kernels (PYRAMID knob 2), with NaN / inf pixels on the corners of every crop, and within
    tolerance of the oracle (which calls scipy.ndimage.zoom like the reference)."""
    from pysilent_amd.util.zoom.from_image import reference_levels
    levels = reference_levels(shape[:2], center, scale)
    frames = np.stack([noise_frame(80 + s_, *shape) for s_ in range(B)])
    frames[0, shape[0] // 2, shape[1] // 2, 1] = np.nan          # inside every crop
    frames[B - 1, shape[0] // 2 - 3, shape[1] // 2 + 5, 0] = np.inf
    for k, (y0, x0, ch, cw) in enumerate(l[:4] for l in levels):  # the corners and edges of every crop: border outputs and mirrored taps
        frames[0, y0, x0, k % 3] = np.inf
        frames[0, y0 + ch - 1, x0 + cw - 1, (k + 1) % 3] = np.nan
        frames[B - 1, y0 + ch // 2, x0, (k + 2) % 3] = -np.inf
        frames[B - 1, y0 + ch - 1, x0 + cw // 3, k % 3] = np.nan
    plan = rt.PyramidPlan(shape[0], shape[1], 3, levels)
    n_plans, px = plan.walk_plans
    assert n_plans == (2 if len(levels) >= 3 else len(levels)) and px in (32, 36), (n_plans, px)
    with rt.tuning(TUNE_PYRAMID, 4):
        per_level = rt.PyramidPlan(shape[0], shape[1], 3, levels)
    assert per_level.walk_plans[0] == len(levels)
    got = plan.run(frames)
    with rt.tuning(TUNE_PYRAMID, 2):
        two = plan.run(frames)
    with rt.tuning(TUNE_PYRAMID, 8):          # the border pixels as a launch of their own instead of the walk launch's first blocks
        own = plan.run(frames)
    a, b, c, d = np.asarray(got.data), np.asarray(two.data), np.asarray(per_level.run(frames).data), np.asarray(own.data)
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.isnan(a), np.isnan(c)) and np.array_equal(np.isnan(a), np.isnan(d))
    np.testing.assert_array_equal(np.nan_to_num(a, nan=7.0), np.nan_to_num(b, nan=7.0))
    np.testing.assert_array_equal(np.nan_to_num(a, nan=7.0), np.nan_to_num(c, nan=7.0))
    np.testing.assert_array_equal(np.nan_to_num(a, nan=7.0), np.nan_to_num(d, nan=7.0))
    with np.errstate(invalid="ignore", over="ignore"):
        bad_want = so.zoom_from_image(frames[0], 3, center, scale)
        assert_close(a.reshape((B,) + bad_want.shape)[0], bad_want, RTOL, scale=255.0, what="reference layout, non-finite pixels on the crop corners",
                     bound=eb.zoom(bad_want))
    clean = noise_frame(98, *shape)
    want = so.zoom_from_image(clean, 3, center, scale)
    res = plan.run(clean[None])
    assert_close(res.data.reshape(want.shape), want, RTOL, scale=255.0, what="reference layout through the walk", bound=eb.zoom(want))


@pytest.mark.parametrize("shape,scale,n", [((270, 480, 3), math.e ** .5, 5), ((135, 240, 3), 1.7, 4), ((200, 300, 3), 1.6, 3)])
def test_rgb_pyramid_walk_takes_zoom_steps_down_to_1_6(rt, shape, scale, n):
    """Classic pyramids at the reference's zoom ratio e ** .5 (and down to 1.6): 32 instead of 36 pixels per consumer wave keep
    the outputs per wave tile within the 21 the gather takes; bit-identical to the unit + region kernels."""
    from pysilent_amd.util.zoom.from_image import classic_levels
    frames = np.stack([noise_frame(90 + s_, *shape) for s_ in range(2)])
    plan = rt.PyramidPlan(shape[0], shape[1], 3, classic_levels(shape[:2], scale, n))
    assert plan.walk_plans == (1, 32)
    got = plan.run(frames)
    with rt.tuning(TUNE_PYRAMID, 2):
        two = plan.run(frames)
    np.testing.assert_array_equal(got.data, two.data)
    want = so.classic_pyramid(frames[1], scale, n)
    for l in range(n):
        assert_close(got.level(l)[1:2], want[l], RTOL, scale=255.0, what="walk px 32 level %d" % l, bound=eb.zoom(want[l]))


@pytest.mark.parametrize("shape,scale,n,px", [((270, 480, 3), 2 ** .5, 6, 28),          # sqrt 2
                                              ((270, 480, 3), 2 ** .5, 10, 0),          # 9 general levels: unit + region kernels
                                              ((200, 300, 3), 2 ** (1 / 3), 5, 24),
                                              ((240, 320, 3), 1.2, 8, 24),
                                              ((1080, 1920, 3), 2 ** .5, 8, 28)])
def test_rgb_pyramid_walk_takes_zoom_steps_down_to_1_2(rt, shape, scale, n, px):
    """Round 5: zoom steps below 1.6 keep the single-read walk -- 28 pixels per consumer wave down to 1.4, 24
    down to 1.2 (a ladder of more than 7 general levels stays on the unit + region kernels: measured faster there).  Bit-identical to the unit + region kernels, non-finite pixels included; within tolerance of the oracle."""
    from pysilent_amd.util.zoom.from_image import classic_levels
    frames = np.stack([noise_frame(120 + s_, *shape) for s_ in range(2)])
    frames[0, shape[0] // 2, shape[1] // 3, 1] = np.nan
    frames[0, 0, shape[1] - 1, 2] = np.inf
    plan = rt.PyramidPlan(shape[0], shape[1], 3, classic_levels(shape[:2], scale, n))
    assert plan.walk_plans == ((1, px) if px else (0, 0))
    got = plan.run(frames)
    with rt.tuning(TUNE_PYRAMID, 2):
        two = plan.run(frames)
    a, b = np.asarray(got.data), np.asarray(two.data)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    np.testing.assert_array_equal(np.nan_to_num(a, nan=7.0), np.nan_to_num(b, nan=7.0))
    want = so.classic_pyramid(frames[1], scale, n)
    for l in range(n):
        assert_close(got.level(l)[1:2], want[l], RTOL, scale=255.0, what="walk px %d level %d" % (px, l), bound=eb.zoom(want[l]))


def test_workspace_follows_the_callers_stream(rt):
    """One workspace per context: a caller that alternates between two streams gets correct results (the library drains
    the previous stream before the workspace changes hands)."""
    import torch
    from pysilent_amd.util.selection import top_value_points
    x = torch.from_numpy(np.stack([noise_frame(s, 64, 96, 3) for s in range(4)])).cuda()
    want = so.top_value_points(x.cpu().numpy(), 0.3)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for i in range(6):
        with torch.cuda.stream(s1 if i % 2 == 0 else s2):
            outs.append(top_value_points(x, 0.3))
    torch.cuda.synchronize()
    for o in outs:
        np.testing.assert_array_equal(o.cpu().numpy(), want)
