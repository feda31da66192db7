"""CPU ORACLE (NumPy) for the pySILEnT line-end hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this file.  The product (``pysilent_amd``) never does: its filters run on the HIP
library or fail loudly.

What it restates (all citations relative to /root/reference):
  * conv2d SAME cross-correlation NHWC x HWIO, stride 1  -- the ``tf.nn.conv2d`` call sites
    slam_recognition/filters/rgc.py:14, filters/rgby.py:11, filters/orientation.py:25,
    util/apply_filter.py:6 (TensorFlow 1.12-1.15 semantics; TF is an unpinned third-party
    dependency, requirements.txt:2,5, and is absent from the image)
  * ``regulate_tensor``      slam_recognition/util/regulator/gaussian_regulator_tensor.py:34-36
  * ``pad_inwards``          slam_recognition/util/selection/isolate_rectangle.py:19-23
  * ``get_value_from_color`` slam_recognition/util/color/get_value.py:6-12
  * 3x3 non-max suppression  slam_recognition/_experimental/vision_filter.py:88-89 (product form)
                             slam_recognition/util/energy/boosting.py:18-22 (fired-mask form)
  * ``top_value_points``     slam_recognition/util/selection/top_value_points.py:8-29
  * ``max_value_indices_region``  .../top_value_points.py:32-45
  * the zoom pyramid         slam_recognition/util/zoom/from_image.py:43-69, calling the SAME
    third-party ``scipy.ndimage.zoom(plane, z, prefilter=False, order=5)`` the reference calls
    (from_image.py:55-59; scipy unpinned in requirements.txt:3, 1.15.3 in this image), plus an
    independent restatement of that resampler (``spline5_zoom``) which is what the C oracle
    and the HIP kernel implement.
  * the op order of the chain  slam_recognition/recognition_testing.py:69-90
  *  ``get_centroids`` util/centroids.py:21-46 with ``index_tensor.from_shape`` util/index_tensor.py:7-20;
    ``get_boosting`` / ``generate_recovery`` util/energy/boosting.py:6-42, util/energy/recovery.py:4-22;
    the rest of ``LineEndDisplayer.compile`` recognition_testing.py:77-100 (``line_end_displayer_tail``)

PINNING STATUS.  The constant kernels are pinned by the reference's own generators
(tests/golden/kernels.npz, produced by tests/golden/make_golden.py importing the reference)
and by the literals of the reference's tests.  The reference holds NO test, fixture or golden
vector for any per-frame op and TensorFlow cannot run here, so the
per-frame ops are **parity unpinned** against the reference itself; they are pinned instead by
(i) analytic known-answer tests, (ii) an independent implementation of the same convolution
(torch.nn.functional.conv2d on CPU), and (iii) scipy.ndimage.zoom itself for the pyramid.

Numerics: activations are stored float32 between ops exactly like the TF graph does; inside one
op the accumulation is float64 and rounded once to float32 (TF accumulates in float32 in an
unspecified order, so the oracle is the centre of the tolerance band, not one edge of it).
"""
import math

import numpy as np

F32 = np.float32


# ----------------------------------------------------------------------------- conv / pointwise

def relu_tf(x):
    """tf.maximum(x, [0]) with Eigen's CPU functor ``(x < 0) ? 0 : x`` -- a NaN stays a NaN."""
    return np.where(x < 0, F32(0), x).astype(F32)


def clip_tf(x, lo, hi):
    """tf.clip_by_value = minimum(maximum(x, lo), hi) with the same NaN-preserving functors."""
    y = np.where(x < F32(lo), F32(lo), x)
    y = np.where(y > F32(hi), F32(hi), y)
    return y.astype(F32)


def conv2d_same(x, k, relu=False, clip_hi=None):
    """out[n,y,x,o] = sum_{dy,dx,i} in[n, y+dy-ph, x+dx-pw, i] * K[dy,dx,i,o]; zero outside.

    SAME at stride 1: pad_total = k-1, pad_before = (k-1)//2.  ``x`` is NHWC float32,
    ``k`` HWIO (rounded to float32 first, like ``tf.constant(k, dtype=tf.float32)``).
    """
    x = np.asarray(x, dtype=F32)
    k32 = np.asarray(k, dtype=F32)
    n, h, w, ci = x.shape
    kh, kw, ki, co = k32.shape
    assert ki == ci, "kernel C_in %d != tensor channels %d" % (ki, ci)
    ph, pw = (kh - 1) // 2, (kw - 1) // 2
    xp = np.zeros((n, h + kh - 1, w + kw - 1, ci), dtype=np.float64)
    xp[:, ph:ph + h, pw:pw + w, :] = x
    k64 = k32.astype(np.float64)
    acc = np.zeros((n, h, w, co), dtype=np.float64)
    for dy in range(kh):
        for dx in range(kw):
            acc += xp[:, dy:dy + h, dx:dx + w, :] @ k64[dy, dx]
    out = acc.astype(F32)
    if relu:
        out = relu_tf(out)
    if clip_hi is not None:
        out = clip_tf(out, 0.0, clip_hi)
    return out


def regulate(x, blur, regulation_value, regulation_root=0.5, flat_policy="ieee"):
    """y = x * (rv / pow(min(conv(x, blur), 1), root)), each op rounded to float32 like the TF graph.

    flat_policy "ieee": literal IEEE replication, 0 * (rv / 0) = NaN on an all-zero window
    (hard part 3).  "zero": y = 0 wherever x == 0.
    """
    x = np.asarray(x, dtype=F32)
    b = conv2d_same(x, blur)
    m = np.where(b > F32(1), F32(1), b).astype(F32)          # tf.minimum(b, [1])
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        p = np.power(m.astype(np.float64), np.float64(F32(regulation_root))).astype(F32)
        r = (F32(regulation_value) / p).astype(F32)
        y = (x * r).astype(F32)
    if flat_policy == "zero":
        y = np.where(x == 0, F32(0), y).astype(F32)
    elif flat_policy != "ieee":
        raise ValueError("flat_policy must be 'ieee' or 'zero'")
    return y


def pad_inwards(x, paddings):
    """tensor * pad(ones(shape - sum(paddings)), paddings): zero a border, by MULTIPLICATION."""
    x = np.asarray(x, dtype=F32)
    mask = np.zeros(x.shape, dtype=F32)
    # paddings that use up an axis leave nothing (the reference cannot even build that graph: tf.ones of a negative
    # extent, isolate_rectangle.py:20; a ragged pyramid's smallest levels get there, and are zeroed)
    sl = tuple(slice(int(a), max(int(a), x.shape[d] - int(b))) for d, (a, b) in enumerate(paddings))
    mask[sl] = 1
    with np.errstate(invalid="ignore"):
        return (mask * x).astype(F32)


def value_from_color(x):
    """reduce_sum over channels (left to right, float32) times float32(1/C)."""
    x = np.asarray(x, dtype=F32)
    c = x.shape[-1]
    s = x[..., 0].copy()
    for i in range(1, c):
        s = (s + x[..., i]).astype(F32)
    inv = F32(1.0) / F32(c)
    return (s * inv).astype(F32)[..., None]


def bw_from_color(x):
    """get_bw_from_color, util/color/get_bw.py:6-13: tensordot with ones (left-to-right float32 sum), then
    where(sum != 0, 1, 0); NaN != 0 is true."""
    x = np.asarray(x, dtype=F32)
    s = x[..., 0].copy()
    with np.errstate(invalid="ignore"):          # inf + (-inf) = NaN is part of the semantics
        for i in range(1, x.shape[-1]):
            s = (s + x[..., i]).astype(F32)
    return np.where(s != 0, F32(1), F32(0)).astype(F32)[..., None]


POOL_LOWEST = np.finfo(F32).min


def pool_max(m, v):
    """One step of tf.nn.max_pool as the reference's device path evaluates it: the reference pins its graph to
    '/device:GPU:0' (recognition_testing.py:64), where TF 1.x runs either its own kernel
    (tensorflow/core/kernels/maxpooling_op_gpu.cu.cc, MaxPoolForwardNHWC: ``maxval = lowest(); if (x > maxval)
    maxval = x``) or cuDNN with CUDNN_NOT_PROPAGATE_NAN -- the default, maxpooling_op.cc reads
    TF_ENABLE_MAXPOOL_NANPROP = false.  So: a NaN never wins, the result does not depend on the order of the taps,
    and a window without a single value above lowest() (all NaN, all -inf, or empty) yields lowest() = -FLT_MAX.
    ``m`` is the running maximum (never NaN), ``v`` the next tap."""
    with np.errstate(invalid="ignore"):
        return np.where(v > m, v, m).astype(F32)


def pool_max_reduce(v, axis):
    """max_pool over whole axes with the same rule (np.fmax ignores NaN; lowest() is the identity)."""
    v = np.asarray(v, dtype=F32)
    with np.errstate(invalid="ignore"):
        m = np.fmax.reduce(np.where(np.isnan(v), POOL_LOWEST, v), axis=axis, initial=POOL_LOWEST)
    return np.asarray(m, dtype=F32)


def maxpool3x3_same(x):
    """tf.nn.max_pool 3x3 stride 1 SAME: out-of-image taps are ignored, NaN taps too (see pool_max)."""
    x = np.asarray(x, dtype=F32)
    n, h, w, c = x.shape
    xp = np.full((n, h + 2, w + 2, c), POOL_LOWEST, dtype=F32)
    xp[:, 1:-1, 1:-1, :] = x
    m = np.full(x.shape, POOL_LOWEST, dtype=F32)
    for dy in range(3):
        for dx in range(3):
            m = pool_max(m, xp[:, dy:dy + h, dx:dx + w, :])
    return m


def nms3x3(x, mode="product"):
    """mode "product": x * where(x == maxpool(x), x, 0)  (== x^2 at maxima; vision_filter.py:88-89)
    mode "fired":   where(x == maxpool(x), 1, 0)       (boosting.py:18-22)."""
    x = np.asarray(x, dtype=F32)
    m = maxpool3x3_same(x)
    is_max = x == m
    if mode == "fired":
        return is_max.astype(F32)
    if mode != "product":
        raise ValueError("mode must be 'product' or 'fired'")
    with np.errstate(invalid="ignore"):
        return (x * np.where(is_max, x, F32(0))).astype(F32)


def level_max_min(v):
    """Per batch item global max and min as top_value_points.py:16-21 computes them: max_pool(v) and
    -1.0 * max_pool(-v) with k = stride = (H, W); NaNs are ignored by both (see pool_max), a level without any
    finite value gives (-FLT_MAX, +FLT_MAX)."""
    v = np.asarray(v, dtype=F32)
    mx = pool_max_reduce(v, (1, 2, 3))
    mn = (F32(-1.0) * pool_max_reduce(-v, (1, 2, 3))).astype(F32)
    return mx, mn


def top_value_points(color, top_percent=0.1, value=None):
    """color * (value >= (1-p)*max + p*min), max/min per batch item; every op float32, unfused."""
    color = np.asarray(color, dtype=F32)
    if value is None:
        value = value_from_color(color)
    value = np.asarray(value, dtype=F32)
    mx, mn = level_max_min(value)
    a = F32(1.0 - top_percent)
    b = F32(top_percent)
    thr = ((a * mx).astype(F32) + (b * mn).astype(F32)).astype(F32)
    with np.errstate(invalid="ignore"):
        mask = (value >= thr[:, None, None, None]).astype(F32)      # a NaN value compares false ...
        return (color * mask).astype(F32)                           # ... and NaN * 0 stays NaN in the colour map


def _region_pool_geometry(size, stride):
    """TF1 max_pool SAME with window == full extent ``size`` and the given stride, then the
    NEAREST resize back to ``size``: returns (n_windows, [(lo, hi)] per window, src index per pixel)."""
    out = -(-size // stride)
    pad_total = max((out - 1) * stride + size - size, 0)
    pad_before = pad_total // 2
    wins = []
    for j in range(out):
        lo = j * stride - pad_before
        wins.append((max(lo, 0), min(lo + size, size)))
    scale = F32(out) / F32(size)                                   # CalculateResizeScale, float32
    src = np.minimum(np.floor((np.arange(size, dtype=F32) * scale).astype(F32)).astype(np.int64), out - 1)
    return out, wins, src


def region_threshold(value, region_h, region_w):
    """The ``resized_pool`` map of max_value_indices_region for one NHW1 tensor."""
    value = np.asarray(value, dtype=F32)
    n, h, w, c = value.shape
    assert c == 1
    oh, wins_y, src_y = _region_pool_geometry(h, int(region_h))
    ow, wins_x, src_x = _region_pool_geometry(w, int(region_w))
    pooled = np.empty((n, oh, ow), dtype=F32)
    for j, (y0, y1) in enumerate(wins_y):
        for i, (x0, x1) in enumerate(wins_x):
            pooled[:, j, i] = pool_max_reduce(value[:, y0:y1, x0:x1, 0], (1, 2))
    return pooled[:, src_y][:, :, src_x][..., None]


def max_value_indices_region(color, region_shape, value=None):
    """int64 [K,4] rows (n, y, x, 0), row-major sorted == tf.where(value >= resized_pool)."""
    if value is None:
        value = value_from_color(color)
    value = np.asarray(value, dtype=F32)
    thr = region_threshold(value, int(region_shape[1]), int(region_shape[2]))
    with np.errstate(invalid="ignore"):
        return np.argwhere(value >= thr).astype(np.int64)           # NaN >= thr is false: a NaN pixel is never a keypoint


# ----------------------------------------------------------------------------- centroids

def index_tensor_from_shape(shape):
    """slam_recognition/util/index_tensor.py:7-20 with are_dimensions_reversed = True: [h, w, 2] holding (x, y).
    Pinned by the reference's tests/test_index_tensor.py:10-11 ([[[0,0],[1,0]],[[0,1],[1,1]]] for 2 x 2)."""
    h, w = int(shape[1]), int(shape[2])
    idx = np.empty((h, w, 2), dtype=np.int32)
    idx[..., 0] = np.arange(w)[None, :]
    idx[..., 1] = np.arange(h)[:, None]
    return idx


def _strided_same_geometry(size, k):
    """tf.nn.convolution(strides = k, window = k, SAME): (n_out, first input index of window 0)."""
    out = -(-size // k)
    pad_total = max((out - 1) * k + k - size, 0)
    return out, -(pad_total // 2)


def get_centroids(value, region_shape):
    """slam_recognition/util/centroids.py:21-46.  value: [N, h, w, 1]; region_shape = [1, rh, rw].
    Returns (value_centroids [N, h, w, 1], total_pool [N, ceil(h/rh), ceil(w/rw), 1]):
      centroid_pool = box sums (stride = window = region, SAME) of (x * v, y * v); total_pool = box sums of v
      (the filter ones/num_channels applied to v tiled to 2 channels); corrected = centroid / total (0/0 = NaN);
      nearest-neighbour resize back; value_centroids = |cx - x| + |cy - y|."""
    value = np.asarray(value, dtype=F32)
    n, h, w, c = value.shape
    assert c == 1
    rh, rw = int(region_shape[1]), int(region_shape[2])
    oh, y_first = _strided_same_geometry(h, rh)
    ow, x_first = _strided_same_geometry(w, rw)
    idx = index_tensor_from_shape((n, h, w, 1)).astype(F32)
    v = value[..., 0]
    bx = (idx[None, :, :, 0] * v).astype(F32)          # ind_tens * full_channel_values, float32
    by = (idx[None, :, :, 1] * v).astype(F32)
    half = (v * F32(0.5)).astype(F32)                  # each of the 2 tiled channels times the 1/2 filter tap
    cx = np.zeros((n, oh, ow), np.float64)
    cy = np.zeros((n, oh, ow), np.float64)
    tot = np.zeros((n, oh, ow), np.float64)
    for j in range(oh):
        y0, y1 = max(y_first + j * rh, 0), min(y_first + j * rh + rh, h)
        for i in range(ow):
            x0, x1 = max(x_first + i * rw, 0), min(x_first + i * rw + rw, w)
            cx[:, j, i] = bx[:, y0:y1, x0:x1].astype(np.float64).sum(axis=(1, 2))
            cy[:, j, i] = by[:, y0:y1, x0:x1].astype(np.float64).sum(axis=(1, 2))
            tot[:, j, i] = 2.0 * half[:, y0:y1, x0:x1].astype(np.float64).sum(axis=(1, 2))
    cx, cy, tot = cx.astype(F32), cy.astype(F32), tot.astype(F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        ccx, ccy = (cx / tot).astype(F32), (cy / tot).astype(F32)
    sy = np.minimum(np.floor((np.arange(h, dtype=F32) * (F32(oh) / F32(h))).astype(F32)).astype(np.int64), oh - 1)
    sx = np.minimum(np.floor((np.arange(w, dtype=F32) * (F32(ow) / F32(w))).astype(F32)).astype(np.int64), ow - 1)
    rx, ry = ccx[:, sy][:, :, sx], ccy[:, sy][:, :, sx]
    with np.errstate(invalid="ignore"):
        dist = (np.abs(rx - idx[None, :, :, 0]).astype(F32) + np.abs(ry - idx[None, :, :, 1]).astype(F32)).astype(F32)
    return dist[..., None], tot[..., None]


# ----------------------------------------------------------------------------- pyramid

def _spline5_weights(t):
    """Quintic cardinal B-spline weights for taps floor(c)-2 .. floor(c)+3 at fraction t in [0,1).

    Same construction as scipy's ni_splines.c: closed forms for five taps, the last one
    by partition of unity.
    """
    y = t
    z = 1.0 - t
    w = [0.0] * 6
    t2 = y * y
    w[2] = t2 * (t2 * (0.25 - y / 12.0) - 0.5) + 0.55
    t2 = z * z
    w[3] = t2 * (t2 * (0.25 - z / 12.0) - 0.5) + 0.55
    y1 = y + 1.0
    w[1] = y1 * (y1 * (y1 * (y1 * (y1 / 24.0 - 0.375) + 1.25) - 1.75) + 0.625) + 0.425
    z1 = z + 1.0
    w[4] = z1 * (z1 * (z1 * (z1 * (z1 / 24.0 - 0.375) + 1.25) - 1.75) + 0.625) + 0.425
    y2 = 1.0 - y                                                    # = 3 - (y + 2), distance-to-support-end
    w[0] = y2 * y2 * y2 * y2 * y2 / 120.0
    w[5] = 1.0 - w[0] - w[1] - w[2] - w[3] - w[4]
    return w


def mirror_index(i, n):
    """scipy 'mirror' extension (d c b | a b c d | c b a): reflect about the edge SAMPLE."""
    if n == 1:
        return 0
    period = 2 * (n - 1)
    i = abs(i) % period
    return period - i if i >= n else i


def zoom_axis_table(n_in, n_out):
    """Per output index: (base = floor(coord), 6 mirrored source indices, 6 float64 weights).

    scipy.ndimage.zoom with grid_mode=False maps output o to o * (n_in-1)/(n_out-1).
    Its default mode 'constant' (the reference passes no mode) declares a coordinate outside
    [0, n_in-1] out of bounds and emits cval = 0 for the whole output row/column; that happens
    when the LAST coordinate lands one ulp above n_in-1 (e.g. 23 * (47/23)).  Encoded here as
    an all-zero weight row.
    """
    step = (n_in - 1) / (n_out - 1) if n_out > 1 else 1.0
    base = np.empty(n_out, dtype=np.int64)
    idx = np.empty((n_out, 6), dtype=np.int64)
    wts = np.empty((n_out, 6), dtype=np.float64)
    for o in range(n_out):
        c = o * step
        b = int(math.floor(c))
        base[o] = b
        wts[o] = _spline5_weights(c - b) if 0.0 <= c <= n_in - 1 else 0.0
        for j in range(6):
            idx[o, j] = mirror_index(b - 2 + j, n_in)
    return base, idx, wts


def zoom_out_size(n_in, zoom):
    """scipy.ndimage.zoom output extent: int(round(n_in * zoom)) with Python's round (half to even)."""
    return int(round(n_in * zoom))


def spline5_zoom(plane, out_h, out_w):
    """Independent restatement of scipy.ndimage.zoom(plane, z, order=5, prefilter=False)
    for a 2-D float32 plane: float64 tap products and sum, one rounding to float32."""
    plane = np.asarray(plane, dtype=F32)
    h, w = plane.shape
    _, iy, wy = zoom_axis_table(h, out_h)
    _, ix, wx = zoom_axis_table(w, out_w)
    p64 = plane.astype(np.float64)
    acc = np.zeros((out_h, out_w), dtype=np.float64)
    for a in range(6):
        rows = p64[iy[:, a], :]
        for b in range(6):
            acc += (wy[:, a][:, None] * wx[:, b][None, :]) * rows[:, ix[:, b]]
    return acc.astype(F32)


def ref_level_geometry(image_hw, center_hw, scale, level):
    """Crop and zoom geometry of one level of image_to_zoom_tensor (from_image.py:49-64).

    Returns (y0, x0, crop_h, crop_w, zoom_h, zoom_w, zoom_factor)."""
    geo = []
    for i, c in zip(image_hw, center_hw):
        sd = c * (scale ** level)
        lo = int(max((i - sd) / 2, 0))
        hi = min(int((i + sd) / 2), i)
        geo.append((lo, hi - lo))
    z = 1.0 / (scale ** level)
    (y0, ch), (x0, cw) = geo
    return y0, x0, ch, cw, zoom_out_size(ch, z), zoom_out_size(cw, z), z


def ref_num_scales(image_hw, center_hw, scale):
    return int(math.ceil(max(math.log(i / c, scale) for i, c in zip(image_hw, center_hw))))


def zoom_from_image(image, num_colors, center_dimensions, scale, use_scipy=True):
    """image_to_zoom_tensor with tuple indexing (the reference's list-of-slices indexing raises
    IndexError on NumPy >= 1.23).  Canvas pixels the zoomed crop does not cover are
    uninitialised in the reference (np.empty, from_image.py:47,53); the oracle defines them as 0.
    Returns float32 [L, h, w, C] (the reference returns float64 holding float32 values)."""
    from scipy import ndimage
    assert scale > 1, "Scale must be greater than one."
    assert num_colors > 0, "Number of colors must be greater than zero."
    for d in center_dimensions:
        assert d > 0, "Each dimension must be larger than zero."
    image = np.asarray(image, dtype=F32)
    image_hw = image.shape[:-1]
    center_hw = list(reversed(center_dimensions))
    n_scales = ref_num_scales(image_hw, center_hw, scale)
    out = np.zeros([n_scales] + center_hw + [num_colors], dtype=F32)
    for s in range(n_scales):
        y0, x0, ch, cw, zh, zw, z = ref_level_geometry(image_hw, center_hw, scale, s)
        crop = image[y0:y0 + ch, x0:x0 + cw]
        for c in range(num_colors):
            plane = np.ascontiguousarray(crop[:, :, c])
            if use_scipy:
                zoomed = ndimage.zoom(plane, z, prefilter=False, order=5)
                assert zoomed.shape == (zh, zw), (zoomed.shape, zh, zw)
            else:
                zoomed = spline5_zoom(plane, zh, zw)
            ym, xm = min(center_hw[0], zh), min(center_hw[1], zw)
            out[s, :ym, :xm, c] = zoomed[:ym, :xm]
    return out


def classic_extents(h, w, scale, n_levels):
    """Level extents of the 'classic layout' (SURVEY.md section 8d): whole frame zoomed by scale^-l."""
    return [(zoom_out_size(h, 1.0 / scale ** l), zoom_out_size(w, 1.0 / scale ** l)) for l in range(n_levels)]


def classic_pyramid(image, scale, n_levels, use_scipy=True):
    """List of float32 [1, H_l, W_l, C] levels: the per-level body of from_image.py:49-64 when
    center_dimensions equals the image size (crop == whole frame, zoom factor scale^-l)."""
    from scipy import ndimage
    image = np.asarray(image, dtype=F32)
    h, w, c = image.shape
    levels = []
    for l, (zh, zw) in enumerate(classic_extents(h, w, scale, n_levels)):
        z = 1.0 / scale ** l
        lev = np.empty((1, zh, zw, c), dtype=F32)
        for ch in range(c):
            plane = np.ascontiguousarray(image[:, :, ch])
            lev[0, :, :, ch] = (ndimage.zoom(plane, z, prefilter=False, order=5) if use_scipy
                                else spline5_zoom(plane, zh, zw))
        levels.append(lev)
    return levels


# ----------------------------------------------------------------------------- chains

def gray_line_end_pass(levels, cs_kernel, end_bank, clip_hi=255.0):
    """BASELINE config 1/2/5 chain on each level: CS -> ReLU -> K-orientation end bank -> ReLU -> clip.

    Returns [(cs_map [1,H,W,1], end_maps [1,H,W,K])] per level."""
    out = []
    for lev in levels:
        cs = conv2d_same(lev, cs_kernel, relu=True)
        end = conv2d_same(cs, end_bank, relu=True, clip_hi=clip_hi)
        out.append((cs, end))
    return out


def rgb_line_end_chain(x, kernels, flat_policy="ieee", blur_root=0.1, blur_rv=1.0, clip_hi=255.0, pad=2):
    """The reference chain, recognition_testing.py:69-77, on one NHWC tensor.

    ``kernels`` = dict(rgc, rgby, stripe, blur, end) of HWIO arrays.
    Returns dict(orient, line_end, padded, value)."""
    rgc = conv2d_same(x, kernels["rgc"], relu=True)
    rgby = conv2d_same(rgc, kernels["rgby"], relu=True)
    stripe = conv2d_same(rgby, kernels["stripe"], relu=True)
    orient = regulate(stripe, kernels["blur"], blur_rv, blur_root, flat_policy)
    line_end = conv2d_same(orient, kernels["end"], relu=True, clip_hi=clip_hi)
    padded = pad_inwards(line_end, [[0, 0], [pad, pad], [pad, pad], [0, 0]])
    value = value_from_color(padded)
    return dict(rgc=rgc, rgby=rgby, stripe=stripe, orient=orient, line_end=line_end, padded=padded, value=value)


# ----------------------------------------------------------------------------- boosting state

def generate_recovery(fire_strength, is_input_based=False, is_constant=True, recovery_amount=10.0,
                      recovery_percentage=0.8):
    """slam_recognition/util/energy/recovery.py:4-22 (float32)."""
    fire_strength = np.asarray(fire_strength, np.float32)
    const = np.full_like(fire_strength, np.float32(recovery_amount))            # ones_like * amount
    inp = fire_strength * np.float32(recovery_percentage)
    if is_input_based and not is_constant:
        return inp
    if is_constant and not is_input_based:
        return const
    if is_input_based and is_constant:
        return np.where(inp < const, const, inp).astype(np.float32)              # tf.maximum, Eigen (a < b) ? b : a
    raise ValueError("You must choose a type of recovery")


def boosting_power(x, energy):
    """``input_tensor ** exhaustion_tensor`` (boosting.py:17).  TF evaluates float32 pow with the device's libm
    (glibc powf on CPU, CUDA powf on GPU: not bit-reproducible across devices); the oracle and the HIP kernel
    both use the correctly rounded form: pow in float64, rounded once to float32."""
    with np.errstate(all="ignore"):
        return np.power(np.asarray(x, np.float64), np.asarray(energy, np.float64)).astype(np.float32)


def get_boosting(x, energy, exhaustion_max=1, excitation_max=1, input_based_recovery=False, constant_recovery=True,
                 for_visualizing=False):
    """slam_recognition/util/energy/boosting.py:10-42.  x, energy: [N,h,w,1] float32.
    Returns (has_fired, new_energy) -- the caller stores new_energy as the next state (the reference's
    ``exhaustion_tensor.assign``); with for_visualizing the two maps are the 3-channel display forms."""
    x = np.asarray(x, np.float32)
    energy = np.asarray(energy, np.float32)
    m = boosting_power(x, energy)
    pooled = maxpool3x3_same(m)
    fired = np.where(m == pooled, np.float32(1), np.float32(0)).astype(np.float32)
    fire_strength = fired * x
    exhaustion = fired * np.float32(255.0)
    recovery = generate_recovery(fire_strength, input_based_recovery, constant_recovery)
    with np.errstate(all="ignore"):
        upd = (energy * np.float32(255.0) - exhaustion + recovery) / np.float32(255.0)
    new_energy = clip_tf(upd, np.float32(-exhaustion_max), np.float32(excitation_max))
    if not for_visualizing:
        return fired, new_energy
    normer = np.float32(255.0 / (exhaustion_max + excitation_max))
    centerer = np.float32((float(excitation_max) / (exhaustion_max + excitation_max)) * 255.0)
    fired2 = np.repeat(fired, 3, axis=-1) * x
    vis = np.repeat(new_energy * normer + centerer, 3, axis=-1)
    return fired2.astype(np.float32), vis.astype(np.float32), new_energy


def initialize_boosting(x, initial_multiplier=8):
    """boosting.py:6-7."""
    return np.full(np.shape(x), np.float32(initial_multiplier), np.float32)


# ----------------------------------------------------------------------------- display graph

def resize_nearest_tf1(x, out_h, out_w):
    """tf.image.resize_nearest_neighbor, TF1 default (align_corners=False): src = min(floor(dst * float32(in/out)),
    in - 1).  x: [N, h, w, C]."""
    x = np.asarray(x)
    n, h, w, c = x.shape
    sy = np.minimum(np.floor((np.arange(out_h, dtype=F32) * (F32(h) / F32(out_h))).astype(F32)).astype(np.int64), h - 1)
    sx = np.minimum(np.floor((np.arange(out_w, dtype=F32) * (F32(w) / F32(out_w))).astype(F32)).astype(np.int64), w - 1)
    return x[:, sy][:, :, sx]


def affine_clip(x, mul=1.0, add=0.0, lo=-np.inf, hi=np.inf, post_add=0.0, div=1.0):
    """clip(x * mul / div + add, lo, hi) + post_add in float32, one rounding per operation (the scalar glue ops
    of recognition_testing.py:79-81, :99: ``t / 255.0``, ``clip(t * (255 / 4.0), 1, 256) - 1``, ``255 - t * 255``)."""
    with np.errstate(all="ignore"):
        y = ((np.asarray(x, F32) * F32(mul)).astype(F32) / F32(div)).astype(F32) + F32(add)
        return (clip_tf(y.astype(F32), lo, hi) + F32(post_add)).astype(F32)


def displayer_half_shape(h, w):
    """recognition_testing.py:82: int32(float32(shape) / float32(e ** .5))."""
    root_e = F32(np.e ** .5)
    return int(F32(h) / root_e), int(F32(w) / root_e)


def line_end_displayer_tail(padded, energy, centroid_region=(1, 3, 3)):
    """The part of LineEndDisplayer.compile after pad_inwards, recognition_testing.py:77-87: value -> centroids,
    importances, half-size centroids, boosting.  Returns ([255 - centroids * 255, 255 - centroids2 * 255,
    fired * 255, update_importances], new_energy)."""
    gray = value_from_color(padded)
    centroids, imp = get_centroids(affine_clip(gray, div=255.0), centroid_region)
    imp = affine_clip(imp, 255 / 4.0, 0.0, 1.0, 256.0, -1.0)
    hh, hw = displayer_half_shape(gray.shape[1], gray.shape[2])
    im2 = resize_nearest_tf1(gray, hh, hw)
    centroids2, _ = get_centroids(affine_clip(im2, div=255.0), centroid_region)
    fired3, vis, new_energy = get_boosting(imp, energy, for_visualizing=True)
    return [affine_clip(centroids, -255.0, 255.0), affine_clip(centroids2, -255.0, 255.0),
            affine_clip(fired3, 255.0), vis], new_energy


def line_end_displayer_run(pyramid, energy, kernels, centroid_region=(1, 3, 3)):
    """LineEndDisplayer.compile + run, recognition_testing.py:60-100, :132: the six fetched tensors and the
    new boosting state.  pyramid: [L, h, w, 3]; energy: [L, ceil(h/3), ceil(w/3), 1]."""
    ch = rgb_line_end_chain(pyramid, kernels)
    tail, new_energy = line_end_displayer_tail(ch["padded"], energy, centroid_region)
    return [ch["orient"]] + tail + [ch["padded"]], new_energy
