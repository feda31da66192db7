"""The oracle itself: pinned against the reference-generated golden kernels, analytic known answers,
an independent convolution (torch), scipy.ndimage.zoom, and its own C port.  CPU only."""
import math

import numpy as np
import numpy.testing as npt
import pytest

import silent_oracle as so
import c_oracle as co
from conftest import noise_frame, structured_frame


def test_conv_impulse_returns_flipped_kernel(golden):
    k = golden["rgb_2d_end_tensors"]
    x = np.zeros((1, 7, 7, 3), np.float32)
    x[0, 3, 3, 2] = 1.0
    out = so.conv2d_same(x, k)
    k32 = k.astype(np.float32)
    for dy in range(3):
        for dx in range(3):
            npt.assert_array_equal(out[0, 3 - (dy - 1), 3 - (dx - 1)], k32[dy, dx, 2])


def test_conv_constant_image_known_answers(golden):
    # midget_rgc sums to 4/3 - 2/3 = 2/3 per diagonal channel; centre 1.33333, edge -0.0976311
    rgc = golden["midget_rgc_2"]
    npt.assert_allclose(rgc[1, 1, 0, 0], 4.0 / 3.0, rtol=1e-12)
    npt.assert_allclose(rgc[0, 1, 0, 0], -0.0976311, atol=1e-7)
    npt.assert_allclose(rgc[0, 0, 0, 0], -0.0690356, atol=1e-7)
    x = np.full((1, 6, 6, 3), 90.0, np.float32)
    out = so.conv2d_same(x, rgc)
    npt.assert_allclose(out[0, 1:-1, 1:-1], 60.0, rtol=1e-6)
    npt.assert_allclose(out[0, 0, 0, 0], 90.0 * rgc[1:, 1:, 0, 0].sum(), rtol=1e-6)     # SAME zero padding


@pytest.mark.parametrize("ks,ci,co_", [(3, 3, 3), (7, 3, 3), (3, 1, 8), (2, 3, 2), (5, 2, 1)])
def test_conv_against_torch(ks, ci, co_):
    torch = pytest.importorskip("torch")
    F = torch.nn.functional
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((2, 19, 23, ci)) * 50).astype(np.float32)
    k = rng.standard_normal((ks, ks, ci, co_))
    got = so.conv2d_same(x, k)
    kt = torch.from_numpy(k.astype(np.float32)).permute(3, 2, 0, 1).double()
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).double()
    pb = (ks - 1) // 2
    want = F.conv2d(F.pad(xt, (pb, ks - 1 - pb, pb, ks - 1 - pb)), kt).permute(0, 2, 3, 1).float().numpy()
    npt.assert_array_equal(got, want)


def test_relu_and_clip_keep_nan():
    x = np.array([np.nan, -1.0, 2.0, 300.0], np.float32)
    r = so.relu_tf(x)
    assert np.isnan(r[0]) and r[1] == 0 and r[2] == 2
    c = so.clip_tf(x, 0, 255)
    assert np.isnan(c[0]) and c[3] == 255


def test_regulate_flat_region_policies(golden):
    x = np.zeros((1, 12, 12, 3), np.float32)
    x[0, 6, 6] = 4.0
    blur = golden["blur_tensor_2_7"]
    y = so.regulate(x, blur, 1.0, 0.1, "ieee")
    assert np.isnan(y[0, 0, 0, 0])                 # 0 * (1 / pow(0, .1)) = 0 * inf
    assert y[0, 6, 6, 0] == 4.0                    # blur >= 1 -> clamp to 1 -> x * 1
    z = so.regulate(x, blur, 1.0, 0.1, "zero")
    assert not np.isnan(z).any() and z[0, 6, 6, 0] == 4.0 and z[0, 0, 0, 0] == 0.0
    small = (x * np.float32(0.01)).astype(np.float32)
    got = so.regulate(small, blur, 2.0, 0.5, "zero")[0, 6, 6, 0]
    b = np.float32(0.04 * 3)                       # centre weight 1 over 3 input channels
    npt.assert_allclose(got, np.float32(0.04) * (np.float32(2.0) / np.float32(np.sqrt(np.float64(b)))), rtol=1e-6)


def test_pad_value_nms_known_answers():
    x = np.arange(1 * 5 * 6 * 3, dtype=np.float32).reshape(1, 5, 6, 3)
    p = so.pad_inwards(x, [[0, 0], [2, 2], [2, 2], [0, 0]])
    assert p[0, 2, 2:4].tolist() == x[0, 2, 2:4].tolist() and p.sum() == x[0, 2, 2:4].sum()
    # paddings that use up an axis (a ragged pyramid's smallest levels): nothing is left, in both oracles
    for pads in ([[0, 0], [0, 6], [0, 0], [0, 0]], [[0, 0], [3, 3], [1, 1], [0, 0]], [[0, 0], [0, 0], [4, 3], [0, 0]]):
        assert not so.pad_inwards(x, pads).any() and not co.pad_inwards(x, pads).any()
    v = so.value_from_color(x)
    npt.assert_array_equal(v[..., 0], ((x[..., 0] + x[..., 1]) + x[..., 2]) * (np.float32(1) / np.float32(3)))
    # get_bw_from_color (get_bw.py:6-13): 1 where the channel sum is not 0; cancelling channels give 0, NaN gives 1
    q = np.array([[[[0, 0, 0], [1, -1, 0], [0, 0, 2], [np.nan, 0, 0], [-3, 1, 1], [1e-30, 0, 0]]]], np.float32)
    assert so.bw_from_color(q)[0, 0, :, 0].tolist() == [0, 0, 1, 1, 1, 1]
    m = np.zeros((1, 4, 4, 1), np.float32)
    m[0, 1, 1, 0] = 3.0
    m[0, 3, 3, 0] = 2.0
    out = so.nms3x3(m, "product")
    assert out[0, 1, 1, 0] == 9.0 and out[0, 3, 3, 0] == 4.0 and out.sum() == 13.0      # x^2 at maxima
    fired = so.nms3x3(m, "fired")
    assert fired[0, 1, 1, 0] == 1 and fired[0, 0, 0, 0] == 0 and fired[0, 0, 3, 0] == 1  # zero plateau corner fires


def test_top_value_points_threshold():
    v = np.arange(20, dtype=np.float32).reshape(1, 4, 5, 1)
    c = np.repeat(v, 3, axis=3)
    out = so.top_value_points(c, 0.1, v)            # thr = 0.9 * 19 + 0.1 * 0 = 17.1
    assert (out[..., 0] > 0).sum() == 2 and out[0, 3, 4, 0] == 19 and out[0, 3, 3, 0] == 18


def test_region_geometry_matches_survey_example():
    # region_shape [1,96,144,3] on 192 x 288: 2x2 overlapping windows rows [0,144), [48,192);
    # cols [0,216), [72,288), mapped back to quadrants
    out, wins, src = so._region_pool_geometry(192, 96)
    assert out == 2 and wins == [(0, 144), (48, 192)]
    assert src[:96].tolist() == [0] * 96 and src[96:].tolist() == [1] * 96
    out, wins, src = so._region_pool_geometry(288, 144)
    assert wins == [(0, 216), (72, 288)]
    v = np.zeros((1, 192, 288, 1), np.float32)
    v[0, 10, 10, 0] = 5.0       # only inside window (0,0)
    v[0, 150, 250, 0] = 7.0     # only inside window (1,1)
    idx = so.max_value_indices_region(None, (1, 96, 144, 3), v)
    # quadrant (0,0) compares against 5, quadrant (1,1) against 7; the mixed quadrants see max = 0 -> every pixel
    assert [10, 10] in idx[:, 1:3].tolist() and [150, 250] in idx[:, 1:3].tolist()
    assert len(idx) == 2 + 2 * 96 * 144
    assert (np.diff(idx[:, 1] * 288 + idx[:, 2]) > 0).all()          # row-major sorted


def test_spline_restatement_is_scipy(golden):
    from scipy import ndimage
    rng = np.random.default_rng(3)
    for h, w, z in [(48, 64, 1.0), (48, 64, 0.5), (67, 33, 1 / math.e ** 0.5), (108, 192, 0.25), (9, 9, 2 / 9.0),
                    (31, 57, 0.77)]:
        p = rng.integers(0, 256, (h, w)).astype(np.float32)
        ref = ndimage.zoom(p, z, prefilter=False, order=5)
        assert ref.shape == (so.zoom_out_size(h, z), so.zoom_out_size(w, z))
        npt.assert_array_equal(so.spline5_zoom(p, *ref.shape), ref)
    p = np.zeros((11, 11), np.float32)
    p[5, 5] = 120.0
    npt.assert_allclose(so.spline5_zoom(p, 11, 11)[5, 3:8] * 120, [66, 1716, 4356, 1716, 66], rtol=1e-6)


def test_zoom_from_image_equals_the_reference_wrapper(golden_pyramid):
    """a-1 pinned on the reference ITSELF: the pyramids its own image_to_zoom_tensor (from_image.py:10-69) produced when
    executed unmodified under NumPy < 1.23's list-of-slices indexing (tests/golden/make_golden_pyramid.py) -- level count,
    centre crops clipped to the image, zoom factors, canvas placement -- against the oracle's restatement, bit for bit,
    with SciPy's zoom and with the oracle's own spline (the form the C port and the HIP kernels implement)."""
    images = golden_pyramid["__images__"]
    golden_pyramid = {k: v for k, v in golden_pyramid.items() if not k.startswith("__")}
    assert len(golden_pyramid) >= 5 and len(images) >= 2
    from pysilent_amd.util.zoom import to_image_list          # pure NumPy display glue (to_image_list.py:7-15)
    for name, want_imgs in images.items():
        got = to_image_list(np.clip(golden_pyramid[name][1], 0, 255))
        assert len(got) == len(want_imgs)
        for g, w in zip(got, want_imgs):
            assert g.dtype == np.uint8 and g.shape == w.shape
            npt.assert_array_equal(g, w, err_msg=name)
    for name, (img, want, par) in golden_pyramid.items():
        center, scale = [int(par[0]), int(par[1])], float(par[2])
        for use_scipy in (True, False):
            got = so.zoom_from_image(img, img.shape[2], center, scale, use_scipy=use_scipy)
            assert got.shape == want.shape, name
            assert want.dtype == np.float64 and np.array_equal(want, want.astype(np.float32), equal_nan=True), name   # f64 holding f32 values
            with np.errstate(invalid="ignore", over="ignore"):
                got = so.zoom_from_image(img, img.shape[2], center, scale, use_scipy=use_scipy)
            npt.assert_array_equal(got, want.astype(np.float32), err_msg=name)      # (NaNs equal NaNs: the two non-finite cases too)
        assert so.ref_num_scales(img.shape[:2], [center[1], center[0]], scale) == want.shape[0]


def test_zoom_from_image_matches_reference_geometry():
    # reference default: 480x640x3 -> [2,192,288,3]; SURVEY 8a-1 level counts
    img = noise_frame(0, 480, 640, 3)
    z = so.zoom_from_image(img, 3, (288, 192), math.e ** .5)
    assert z.shape == (2, 192, 288, 3) and z.dtype == np.float32
    npt.assert_array_equal(z, so.zoom_from_image(img, 3, (288, 192), math.e ** .5, use_scipy=False))
    assert so.ref_num_scales((1080, 1920), (192, 288), math.e ** .5) == 4
    assert so.ref_num_scales((2160, 3840), (192, 288), math.e ** .5) == 6
    assert so.classic_extents(1080, 1920, 2.0, 5) == [(1080, 1920), (540, 960), (270, 480), (135, 240), (68, 120)]
    assert sum(h * w for h, w in so.classic_extents(1080, 1920, 2.0, 5)) == 2762160       # P
    assert sum(h * w for h, w in so.classic_extents(2160, 3840, 2.0, 8)) == 11059110


def test_c_port_equals_numpy_oracle(golden, kernels):
    x = structured_frame(1, 41, 57, 3, 25)[None] + noise_frame(1, 41, 57, 3)[None] * np.float32(0.1)
    for name in ("rgc", "rgby", "stripe", "end", "blur"):
        npt.assert_array_equal(co.conv2d_same(x, kernels[name], relu=True), so.conv2d_same(x, kernels[name], relu=True))
    s = so.conv2d_same(x, kernels["stripe"], relu=True)
    for pol in ("ieee", "zero"):
        npt.assert_array_equal(co.regulate(s, kernels["blur"], 1.0, 0.1, pol), so.regulate(s, kernels["blur"], 1.0, 0.1, pol))
    pads = [[0, 0], [2, 2], [2, 2], [0, 0]]
    npt.assert_array_equal(co.pad_inwards(x, pads), so.pad_inwards(x, pads))
    npt.assert_array_equal(co.value_from_color(x), so.value_from_color(x))
    npt.assert_array_equal(co.nms3x3(x), so.nms3x3(x))
    npt.assert_array_equal(co.top_value_points(x, 0.37), so.top_value_points(x, 0.37))
    npt.assert_array_equal(co.max_value_indices_region(x, (1, 20, 28, 3)), so.max_value_indices_region(x, (1, 20, 28, 3)))
    img = noise_frame(2, 97, 131, 1)
    ext = so.classic_extents(97, 131, 1.7, 4)
    for a, b in zip(co.classic_pyramid(img, ext), so.classic_pyramid(img, 1.7, 4)):
        npt.assert_array_equal(a, b)
    cs, end = co.gray_line_end_level(img[None], kernels["cs_gray"], kernels["end4"])
    (wcs, wend), = so.gray_line_end_pass([img[None]], kernels["cs_gray"], kernels["end4"])
    npt.assert_array_equal(cs, wcs)
    npt.assert_array_equal(end, wend)


def test_chain_on_noise_has_no_nan(kernels):
    # SURVEY hard part 3: parity fixtures on noise frames must keep the NaN set empty
    out = so.rgb_line_end_chain(noise_frame(0, 48, 64, 3)[None], kernels)
    for k, v in out.items():
        assert not np.isnan(v).any(), k
    assert out["padded"][0, :2].sum() == 0 and out["value"].shape == (1, 48, 64, 1)


# ----------------------------------------------------------------------------- 8f: centroids, boosting (hand-worked)

def test_centroids_known_answer():
    v = np.zeros((1, 6, 9, 1), np.float32)
    v[0, 1, 4, 0] = 2.0
    v[0, 2, 5, 0] = 2.0
    dist, total = so.get_centroids(v, [1, 3, 3])
    assert total.shape == (1, 2, 3, 1) and dist.shape == v.shape
    npt.assert_array_equal(total[0, :, :, 0], [[0, 4, 0], [0, 0, 0]])
    npt.assert_array_equal(dist[0, :3, 3:6, 0], [[3, 2, 2], [2, 1, 1], [2, 1, 1]])   # centroid (x, y) = (4.5, 1.5)
    assert np.isnan(dist[0, :, :3, 0]).all() and np.isnan(dist[0, 3:, :, 0]).all()    # empty cells: 0/0


def test_centroids_same_padding_geometry():
    # 7 px, window 3, stride 3, SAME: 3 cells, pad_total 2 -> the first cell starts at -1 (covers px 0..1)
    assert so._strided_same_geometry(7, 3) == (3, -1)
    assert so._strided_same_geometry(6, 3) == (2, 0)
    assert so._strided_same_geometry(8, 3) == (3, 0)
    v = np.ones((1, 7, 7, 1), np.float32)
    _, total = so.get_centroids(v, [1, 3, 3])
    npt.assert_array_equal(total[0, :, :, 0], [[4, 6, 4], [6, 9, 6], [4, 6, 4]])


def test_boosting_known_answer():
    x = np.array([[0, 2], [3, 2]], np.float32).reshape(1, 2, 2, 1)
    e = np.ones_like(x)
    fired, new_e = so.get_boosting(x, e)
    npt.assert_array_equal(fired[0, :, :, 0], [[0, 0], [1, 0]])
    # not fired: clip((255 + 10)/255, -1, 1) = 1; fired: (255 - 255 + 10)/255
    npt.assert_allclose(new_e[0, :, :, 0], [[1, 1], [np.float32(10) / np.float32(255), 1]], rtol=0, atol=0)
    f3, vis, st = so.get_boosting(x, e, for_visualizing=True)
    npt.assert_array_equal(st, new_e)
    npt.assert_array_equal(f3[0, 1, 0], [3, 3, 3])
    npt.assert_allclose(vis[0, 0, 0], [255, 255, 255])                                # 1 * 127.5 + 127.5
    both = so.generate_recovery(np.array([0, 20], np.float32), True, True)
    npt.assert_array_equal(both, [10, 16])
    with pytest.raises(ValueError):
        so.generate_recovery(x, False, False)
    assert (so.initialize_boosting(x) == 8).all()


# ----------------------------------------------------------------------------- second opinions for the TF-semantic ops
# (everything below is checked against an independent torch-CPU implementation of the same rule)

def test_maxpool3x3_same_against_torch():
    torch = pytest.importorskip("torch")
    x = noise_frame(3, 23, 31, 3)[None]
    x[0, 5, 7, 1] = -np.inf
    want = torch.nn.functional.max_pool2d(torch.from_numpy(x).permute(0, 3, 1, 2), 3, stride=1, padding=1)   # -inf padding
    npt.assert_array_equal(so.maxpool3x3_same(x), want.permute(0, 2, 3, 1).numpy())
    npt.assert_array_equal(co.nms3x3(x, "fired"), so.nms3x3(x, "fired"))


def test_max_pool_ignores_nan_like_the_tf1_gpu_kernel():
    """TF 1.x on '/device:GPU:0' (where the reference pins its graph): `maxval = lowest(); if (x > maxval) maxval = x`
    and cuDNN NOT_PROPAGATE_NAN -- a NaN never wins, an all-NaN window gives lowest()."""
    lowest = np.finfo(np.float32).min
    x = np.array([[1.0, np.nan, 3.0], [np.nan, np.nan, np.nan], [0.5, 2.0, np.nan]], np.float32)[None, :, :, None]
    m = so.maxpool3x3_same(x)[0, :, :, 0]
    npt.assert_array_equal(m, [[1.0, 3.0, 3.0], [2.0, 3.0, 3.0], [2.0, 2.0, 2.0]])
    allnan = np.full((1, 4, 4, 1), np.nan, np.float32)
    assert (so.maxpool3x3_same(allnan) == lowest).all()
    mx, mn = so.level_max_min(x)
    assert mx[0] == 3.0 and mn[0] == 0.5
    mx, mn = so.level_max_min(allnan)
    assert mx[0] == lowest and mn[0] == -lowest
    # the mask stages: a NaN value is never selected, and NaN * 0 stays NaN in the colour map
    color = np.repeat(x, 3, axis=3)
    top = so.top_value_points(color, 0.5)                     # thr = 0.5 * 3 + 0.5 * 0.5 = 1.75
    assert np.isnan(top[0, 0, 1]).all() and (top[0, 0, 0] == 0).all() and (top[0, 0, 2] == 3).all()
    idx = so.max_value_indices_region(None, (1, 3, 3, 1), x)
    npt.assert_array_equal(idx, [[0, 0, 2, 0]])
    assert len(so.max_value_indices_region(None, (1, 2, 2, 1), allnan)) == 0
    # the C port follows the same rule
    npt.assert_array_equal(co.nms3x3(x), so.nms3x3(x))
    npt.assert_array_equal(co.top_value_points(color, 0.5), top)
    npt.assert_array_equal(co.max_value_indices_region(None, (1, 3, 3, 1), x), idx)


@pytest.mark.parametrize("h,w,oh,ow", [(192, 288, 116, 174), (7, 5, 3, 2), (5, 9, 5, 9), (3, 4, 7, 9)])
def test_resize_nearest_tf1_against_torch(h, w, oh, ow):
    """TF1 ResizeNearestNeighbor (align_corners=False): src = min(floor(dst * float32(in / out)), in - 1) -- the same
    rule as torch's legacy 'nearest' mode."""
    torch = pytest.importorskip("torch")
    x = noise_frame(4, h, w, 2)[None]
    want = torch.nn.functional.interpolate(torch.from_numpy(x).permute(0, 3, 1, 2), size=(oh, ow), mode="nearest")
    npt.assert_array_equal(so.resize_nearest_tf1(x, oh, ow), want.permute(0, 2, 3, 1).numpy())


@pytest.mark.parametrize("h,w,rh,rw", [(192, 288, 96, 144), (37, 53, 18, 26), (16, 16, 5, 7), (9, 31, 9, 4), (40, 40, 3, 3)])
def test_region_pool_and_where_order_against_torch(h, w, rh, rw):
    """max_value_indices_region = max_pool(k = (H, W), stride = region, SAME) > NEAREST resize > tf.where, rebuilt with
    torch: explicit SAME padding (pad_before = pad_total // 2, -inf) + max_pool2d, legacy nearest interpolate, nonzero
    (row-major, like tf.where)."""
    torch = pytest.importorskip("torch")
    F = torch.nn.functional
    v = noise_frame(6, h, w, 1)[None]
    v[0, :3, :3] = 300.0                                       # a tie plateau
    t = torch.from_numpy(v).permute(0, 3, 1, 2)
    oh, ow = -(-h // rh), -(-w // rw)
    ph, pw = max((oh - 1) * rh + h - h, 0), max((ow - 1) * rw + w - w, 0)
    padded = F.pad(t, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2), value=float("-inf"))
    pooled = F.max_pool2d(padded, (h, w), stride=(rh, rw))
    assert pooled.shape[-2:] == (oh, ow)
    thr = F.interpolate(pooled, size=(h, w), mode="nearest")
    want = torch.nonzero((t >= thr).permute(0, 2, 3, 1)).numpy()
    npt.assert_array_equal(so.max_value_indices_region(None, (1, rh, rw, 1), v), want)
    npt.assert_array_equal(co.max_value_indices_region(None, (1, rh, rw, 1), v), want)


@pytest.mark.parametrize("h,w,r", [(192, 288, 3), (37, 53, 3), (10, 11, 4), (5, 5, 2)])
def test_centroid_box_sums_against_torch(h, w, r):
    """get_centroids' strided box sums (tf.nn.convolution, window = stride = region, SAME) rebuilt with torch conv2d on
    explicitly SAME-padded inputs; float64 there, so compare at float32 resolution."""
    torch = pytest.importorskip("torch")
    F = torch.nn.functional
    v = (noise_frame(8, h, w, 1)[None] / np.float32(255.0)).astype(np.float32)
    cent, total = so.get_centroids(v, [1, r, r])
    t = torch.from_numpy(v).permute(0, 3, 1, 2).double()
    oh, ow = -(-h // r), -(-w // r)
    ph, pw = max((oh - 1) * r + r - h, 0), max((ow - 1) * r + r - w, 0)
    pad = (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2)
    ones = torch.ones(1, 1, r, r, dtype=torch.float64)
    tot = F.conv2d(F.pad(t, pad), ones, stride=r)
    xs = torch.arange(w, dtype=torch.float64)[None, None, None, :].expand(1, 1, h, w)
    ys = torch.arange(h, dtype=torch.float64)[None, None, :, None].expand(1, 1, h, w)
    cx = F.conv2d(F.pad(t * xs, pad), ones, stride=r) / tot
    cy = F.conv2d(F.pad(t * ys, pad), ones, stride=r) / tot
    npt.assert_allclose(total[0, :, :, 0], tot[0, 0].numpy(), rtol=2e-6)
    cxr = F.interpolate(cx, size=(h, w), mode="nearest")
    cyr = F.interpolate(cy, size=(h, w), mode="nearest")
    want = ((cxr - xs).abs() + (cyr - ys).abs())[0, 0].numpy()
    npt.assert_allclose(cent[0, :, :, 0], want, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("policy", ["zero", "ieee"])
def test_rgb_chain_composition_against_torch(kernels, policy):
    """Second opinion for the oracle's WHOLE reference graph (recognition_testing.py:69-77, a-2 ... a-8), stage by stage in
    torch float32 ops composed independently of oracle/silent_oracle.py: conv2d (cross-correlation, zero padding), relu,
    the regulator x * (rv / min(conv7x7(x), 1) ** root), relu + clip, the 2-pixel border mask and the channel mean as
    sum * float32(1/3).  The oracle accumulates its convolutions in float64 and rounds once, torch in float32: agreement
    within 2e-5 of each map's range, and the NaN pattern of the 'ieee' policy (0 * inf on flat regions) must be the same
    wherever the blur is exactly 0."""
    torch = pytest.importorskip("torch")
    F = torch.nn.functional
    x = np.stack([noise_frame(3, 40, 56, 3), structured_frame(4, 40, 56, 3, 12)])
    want = so.rgb_line_end_chain(x, kernels, flat_policy=policy)

    def conv(t, k):
        kt = torch.from_numpy(np.asarray(k, np.float64).astype(np.float32)).permute(3, 2, 0, 1)
        p = (kt.shape[-1] - 1) // 2
        return F.conv2d(F.pad(t, (p, p, p, p)), kt)

    t = torch.from_numpy(x).permute(0, 3, 1, 2)
    rgc = torch.relu(conv(t, kernels["rgc"]))
    rgby = torch.relu(conv(rgc, kernels["rgby"]))
    stripe = torch.relu(conv(rgby, kernels["stripe"]))
    blur = conv(stripe, kernels["blur"])
    orient = stripe * (1.0 / torch.pow(torch.clamp(blur, max=1.0), 0.1))
    if policy == "zero":
        orient = torch.where(stripe == 0, torch.zeros_like(orient), orient)
    line = torch.clamp(torch.relu(conv(orient, kernels["end"])), max=255.0)
    mask = torch.zeros_like(line)
    mask[:, :, 2:-2, 2:-2] = 1.0
    padded = line * mask
    value = padded.sum(1, keepdim=True) * np.float32(1.0 / 3.0)
    nhwc = lambda a: a.permute(0, 2, 3, 1).numpy()
    for name, got in (("rgc", rgc), ("rgby", rgby), ("stripe", stripe)):
        g, w = nhwc(got), want[name]
        assert np.abs(g - w).max() <= 2e-5 * max(np.abs(w).max(), 1.0), name
    b = nhwc(blur)
    sure = (b == 0) | (b > 1e-4 * b.max())          # away from float32 rounding residue of the blur around 0
    for name, got in (("orient", orient), ("padded", padded), ("value", value)):
        g, w = nhwc(got), want[name]
        s = sure if g.shape[-1] == 3 else sure.all(axis=-1, keepdims=True)
        if name != "orient":                         # one erosion: a line-end value reads a 3x3 neighbourhood of orient
            e = torch.from_numpy((~sure).astype(np.float32)).permute(0, 3, 1, 2).sum(1, keepdim=True)
            grown = F.max_pool2d(e, 3, 1, 1).permute(0, 2, 3, 1).numpy() > 0
            s = ~grown if g.shape[-1] == 1 else np.broadcast_to(~grown, g.shape)
        assert np.array_equal(np.isnan(g)[s], np.isnan(w)[s]), name + " NaN pattern"
        ok = s & ~np.isnan(w)
        scale = max(np.abs(w[ok]).max(), 1.0) if ok.any() else 1.0
        assert np.abs(g[ok] - w[ok]).max() <= 2e-5 * scale, name
    if policy == "ieee":
        assert np.isnan(want["orient"]).any()       # the line drawing has flat regions: the policy was exercised


@pytest.mark.parametrize("policy", ["ieee", "zero"])
def test_rounding_bounds_hold_for_an_independent_float32_evaluation(kernels, policy):
    """tests/err_bound.py (the element-wise rule of conftest.assert_close): torch's float32 convolutions are a float32
    evaluation of the reference graph in ANOTHER order than the HIP kernels' -- every element of every map must sit inside
    the propagated rounding bound (taps + 4) * 2^-24 * sum |w||x| around the oracle, and the bound must not be vacuous:
    finite on nearly all elements, and on most of them a small multiple of 2^-24 of the largest intermediate response (the
    unclipped stripe / orientation responses of a noise frame reach thousands: that, not the clipped 0..255 result, is the
    scale of the terms that cancel in the line-end bank)."""
    torch = pytest.importorskip("torch")
    import err_bound as eb
    F = torch.nn.functional
    x = np.stack([noise_frame(13, 40, 56, 3), structured_frame(14, 40, 56, 3, 12)])
    want = so.rgb_line_end_chain(x, kernels, flat_policy=policy)
    bound = eb.rgb_chain(x, kernels, want, flat_policy=policy)

    def conv(t, k):
        kt = torch.from_numpy(np.asarray(k, np.float64).astype(np.float32)).permute(3, 2, 0, 1)
        p = (kt.shape[-1] - 1) // 2
        return F.conv2d(F.pad(t, (p, p, p, p)), kt)

    t = torch.from_numpy(x).permute(0, 3, 1, 2)
    got = {}
    got["rgc"] = torch.relu(conv(t, kernels["rgc"]))
    got["rgby"] = torch.relu(conv(got["rgc"], kernels["rgby"]))
    got["stripe"] = torch.relu(conv(got["rgby"], kernels["stripe"]))
    blur = conv(got["stripe"], kernels["blur"])
    orient = got["stripe"] * (1.0 / torch.pow(torch.clamp(blur, max=1.0), 0.1))
    if policy == "zero":
        orient = torch.where(got["stripe"] == 0, torch.zeros_like(orient), orient)
    got["orient"] = orient
    got["line_end"] = torch.clamp(torch.relu(conv(orient, kernels["end"])), max=255.0)
    mask = torch.zeros_like(got["line_end"])
    mask[:, :, 2:-2, 2:-2] = 1.0
    got["padded"] = got["line_end"] * mask
    got["value"] = got["padded"].sum(1, keepdim=True) * np.float32(1.0 / 3.0)
    for name in ("rgc", "rgby", "stripe", "orient", "line_end", "padded", "value"):
        g = got[name].permute(0, 2, 3, 1).numpy().astype(np.float64)
        w, e = want[name].astype(np.float64), bound[name]
        ok = np.isfinite(g) & np.isfinite(w) & ~eb.unbounded(e)
        assert (np.abs(g - w)[ok] <= e[ok] + 1e-38).all(), (name, float((np.abs(g - w)[ok] / np.maximum(e[ok], 1e-300)).max()))
        assert ok.mean() > (0.5 if policy == "ieee" and name != "rgc" and name != "rgby" and name != "stripe" else 0.97), (name, ok.mean())
        terms = max(float(np.abs(v[np.isfinite(v)]).max()) for v in want.values())
        assert np.median(e[ok]) < 400 * 2.0 ** -24 * terms, (name, float(np.median(e[ok])), terms)


def test_rounding_bound_of_a_single_convolution_and_of_the_gray_chain(kernels):
    """The bound for one cancelling stencil: float32 accumulation (torch) against the oracle, and the two-stage gray chain."""
    torch = pytest.importorskip("torch")
    import err_bound as eb
    F = torch.nn.functional
    x = noise_frame(15, 37, 53, 1)[None]
    cs, end = so.gray_line_end_pass([x], kernels["cs_gray"], kernels["end4"])[0]
    e_cs, e_end = eb.gray_chain(x, kernels["cs_gray"], kernels["end4"], cs)

    def conv(t, k):
        kt = torch.from_numpy(np.asarray(k, np.float64).astype(np.float32)).permute(3, 2, 0, 1)
        return F.conv2d(F.pad(t, (1, 1, 1, 1)), kt)

    t = torch.from_numpy(x).permute(0, 3, 1, 2)
    g_cs = torch.relu(conv(t, kernels["cs_gray"]))
    g_end = torch.clamp(torch.relu(conv(g_cs, kernels["end4"])), max=255.0)
    for g, w, e in ((g_cs, cs, e_cs), (g_end, end, e_end)):
        d = np.abs(g.permute(0, 2, 3, 1).numpy().astype(np.float64) - w)
        assert (d <= e + 1e-38).all() and e.max() < 2e-3 and np.median(e) < 1e-3      # (1e-5 * 255 = 2.5e-3)
    # the pyramid's bound is a few ulps of the level itself
    lev = so.classic_pyramid(x[0], 2.0, 2)[1]
    assert np.all(eb.zoom(lev) <= 16 * 2.0 ** -24 * np.abs(lev) + 1e-11)


@pytest.mark.parametrize("policy", ["ieee", "zero"])
def test_c_rgb_pass_frames_equals_the_per_op_composition(kernels, policy):
    """oracle/silent_oracle.c so_rgb_pass_frames (bench.py's CPU baseline for BASELINE config 3: one frame per OpenMP thread,
    pyramid -> reference chain -> top 10 % -> NMS -> value -> per-region indices) gives the rows of the per-op composition the
    GPU parity tests use (tests/kp_margin.py oracle_keypoints), frame by frame, in order."""
    import c_oracle as co
    import kp_margin as km
    from conftest import margin_frame, noise_frame, structured_frame
    h, w, levels = 96, 144, 3
    frames = np.stack([noise_frame(3, h, w, 3), structured_frame(4, h, w, 3, n_lines=20), margin_frame(2, h, w),
                       np.zeros((h, w, 3), np.float32)])
    ks = {k: kernels[k].astype(np.float32) for k in ("rgc", "rgby", "stripe", "blur", "end")}
    ext = so.classic_extents(h, w, 2.0, levels)
    counts, rows = co.rgb_pass_frames(frames, ext, ks, policy, cap=2 * h * w)
    for f in range(len(frames)):
        want, _ = km.oracle_keypoints(frames[f], levels, ks, policy)
        assert counts[f] == len(want)
        np.testing.assert_array_equal(rows[f], want)
    # counting only (what the timed baseline does) gives the same counts
    counts2, none = co.rgb_pass_frames(frames, ext, ks, policy)
    assert none is None and np.array_equal(counts, counts2)
