"""Size-independent properties: hypothesis on the CPU oracle, and the same properties of
the HIP path at BASELINE sizes where the oracle would take too long."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import silent_oracle as so
from conftest import noise_frame, structured_frame

F32 = np.float32


def _img(seed, h, w, c):
    return np.random.default_rng(seed).integers(0, 256, (1, h, w, c)).astype(F32)


# ----------------------------------------------------------------------------- oracle (CPU)

@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 10 ** 6), h=st.integers(3, 20), w=st.integers(3, 20), ci=st.sampled_from([1, 3]),
       co_=st.sampled_from([1, 3, 4]), a=st.floats(-3, 3), b=st.floats(-3, 3))
def test_oracle_conv_is_linear(seed, h, w, ci, co_, a, b):
    rng = np.random.default_rng(seed)
    k = rng.standard_normal((3, 3, ci, co_))
    x, y = _img(seed, h, w, ci), _img(seed + 1, h, w, ci)
    lhs = so.conv2d_same((F32(a) * x + F32(b) * y).astype(F32), k)
    rhs = F32(a) * so.conv2d_same(x, k) + F32(b) * so.conv2d_same(y, k)
    np.testing.assert_allclose(lhs, rhs, rtol=0, atol=2e-3 * (abs(a) + abs(b) + 1))


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 10 ** 6), dy=st.integers(-3, 3), dx=st.integers(-3, 3))
def test_oracle_conv_chain_is_shift_equivariant_away_from_borders(seed, dy, dx):
    rng = np.random.default_rng(seed)
    k1, k2 = rng.standard_normal((3, 3, 1, 1)), rng.standard_normal((3, 3, 1, 4))
    big = _img(seed, 30, 34, 1)
    a = big[:, 4:24, 4:28]
    b = big[:, 4 + dy:24 + dy, 4 + dx:28 + dx]
    fa = so.conv2d_same(so.conv2d_same(a, k1, relu=True), k2, relu=True, clip_hi=255.0)
    fb = so.conv2d_same(so.conv2d_same(b, k1, relu=True), k2, relu=True, clip_hi=255.0)
    # interior of both crops: the same source pixels, 2 px away from every border
    ya, yb = slice(2 + max(dy, 0), 18 + min(dy, 0)), slice(2 + max(dy, 0) - dy, 18 + min(dy, 0) - dy)
    xa, xb = slice(2 + max(dx, 0), 22 + min(dx, 0)), slice(2 + max(dx, 0) - dx, 22 + min(dx, 0) - dx)
    np.testing.assert_array_equal(fa[:, ya, xa], fb[:, yb, xb])


@settings(max_examples=20, deadline=None)
@given(n_in=st.integers(6, 60), frac=st.floats(0.2, 1.0), c=st.floats(-100, 100))
def test_oracle_spline_weights_sum_to_one(n_in, frac, c):
    """Quintic B-spline taps are a partition of unity: a constant row stays constant (except where SciPy's
    mode-'constant' artefact zeroes the last sample, which the restatement reproduces)."""
    n_out = max(2, int(round(n_in * frac)))
    base, idx, w = so.zoom_axis_table(n_in, n_out)
    sums = np.asarray(w, np.float64).sum(axis=1)
    assert np.all((np.abs(sums - 1.0) < 1e-12) | (sums == 0.0))
    assert (sums[:-1] != 0.0).all()


@settings(max_examples=20, deadline=None)
@given(seed=st.integers(0, 10 ** 6), p1=st.floats(0.0, 1.0), p2=st.floats(0.0, 1.0))
def test_oracle_top_percent_is_monotone(seed, p1, p2):
    lo, hi = min(p1, p2), max(p1, p2)
    x = _img(seed, 12, 17, 3)
    v = so.value_from_color(x)
    keep_lo = so.top_value_points(x, lo, v) != 0
    keep_hi = so.top_value_points(x, hi, v) != 0
    assert not (keep_lo & ~keep_hi).any()          # a larger percentage keeps a superset


@settings(max_examples=20, deadline=None)
@given(seed=st.integers(0, 10 ** 6), h=st.integers(2, 24), w=st.integers(2, 24), rh=st.integers(1, 24), rw=st.integers(1, 24))
def test_oracle_keypoints_are_sorted_unique_and_contain_the_global_maximum(seed, h, w, rh, rw):
    v = np.random.default_rng(seed).integers(0, 8, (2, h, w, 1)).astype(F32)
    idx = so.max_value_indices_region(None, (1, rh, rw, 3), v)
    assert idx.dtype == np.int64 and idx.shape[1] == 4
    keys = idx[:, 0] * (h * w) + idx[:, 1] * w + idx[:, 2]
    assert (np.diff(keys) > 0).all()               # row-major, strictly increasing: sorted and unique
    for n in range(2):
        ys, xs = np.nonzero(v[n, :, :, 0] == v[n].max())
        got = {(int(a), int(b)) for a, b in idx[idx[:, 0] == n][:, 1:3]}
        assert {(int(a), int(b)) for a, b in zip(ys, xs)} <= got


# ----------------------------------------------------------------------------- HIP path at full size

@pytest.fixture(scope="module")
def rt():
    from pysilent_amd import _runtime
    return _runtime


@pytest.mark.gpu
def test_gpu_filters_are_shift_equivariant_at_1080p(rt, kernels):
    """Bit-identical responses for the same source pixels at two different positions in the frame (away from the
    borders): tile boundaries, wave halos and the row/column bookkeeping of the streaming kernels cancel out."""
    big = noise_frame(11, 1080 + 37, 1920 + 61, 1)[None]
    a, b = big[:, :1080, :1920], big[:, 37:, 61:]
    ca, ea = rt.gray_line_end(np.ascontiguousarray(a), kernels["cs_gray"], kernels["end4"])
    cb, eb = rt.gray_line_end(np.ascontiguousarray(b), kernels["cs_gray"], kernels["end4"])
    np.testing.assert_array_equal(ca[:, 39:-2, 63:-2], cb[:, 2:-39, 2:-63])
    np.testing.assert_array_equal(ea[:, 39:-2, 63:-2], eb[:, 2:-39, 2:-63])


@pytest.mark.gpu
def test_gpu_pass_is_positively_homogeneous_below_the_clip(rt, kernels):
    """pyramid, CS and line-end responses scale exactly with a power-of-two gain (every operation is linear or a
    ReLU); the clip at 255 is kept out of reach by the gain."""
    from pysilent_amd.util.zoom.from_image import classic_levels
    frame = (structured_frame(3, 1080, 1920, 1) * F32(1 / 64)).astype(F32)
    plan = rt.PyramidPlan(1080, 1920, 1, classic_levels((1080, 1920), 2.0, 5))
    p1, c1, e1 = plan.gray_pass(frame[None], kernels["cs_gray"], kernels["end4"])
    p2, c2, e2 = plan.gray_pass((frame * F32(0.25))[None], kernels["cs_gray"], kernels["end4"])
    assert float(e1.data.max()) < 255.0
    np.testing.assert_array_equal(p2.data, p1.data * F32(0.25))
    np.testing.assert_array_equal(c2.data, c1.data * F32(0.25))
    np.testing.assert_array_equal(e2.data, e1.data * F32(0.25))


@pytest.mark.gpu
def test_gpu_keypoints_sorted_and_selection_monotone_at_1080p(rt, kernels):
    from pysilent_amd.util.selection import max_value_indices_region, top_value_points
    from pysilent_amd.util.color import get_value_from_color
    rng = np.random.default_rng(21)
    x = np.floor(rng.random((2, 1080, 1920, 3)) * 64).astype(F32)
    v = get_value_from_color(x)
    kp = max_value_indices_region(x, [1, 540, 960, 3], v)          # [K, 4] rows (n, y, x, 0) like tf.where
    all_keys = (kp[:, 0] * 1080 + kp[:, 1]) * 1920 + kp[:, 2]
    assert (np.diff(all_keys) > 0).all() and (kp[:, 3] == 0).all()
    for n in range(2):
        idx = kp[kp[:, 0] == n]
        assert len(idx) > 0
        ys, xs = np.nonzero(v[n, :, :, 0] == v[n].max())
        assert {(int(a), int(b)) for a, b in zip(ys, xs)} <= {(int(a), int(b)) for a, b in idx[:, 1:3]}
    keep10 = top_value_points(x, 0.1, v) != 0
    keep30 = top_value_points(x, 0.3, v) != 0
    assert not (keep10 & ~keep30).any() and keep30.sum() > keep10.sum()


@pytest.mark.gpu
def test_random_cases_through_the_c_abi():
    """A fixed-seed slice of scripts/fuzz_gpu.py (random extents, zoom steps, level counts, banks, batch sizes, frame kinds and
    development knobs against the oracle): 40 cases of each kind."""
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_gpu", os.path.join(root, "scripts", "fuzz_gpu.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    k = fz.make_kernels()
    from pysilent_amd import _runtime
    ctx = _runtime.get_context()
    saved = [ctx.get_tuning(i) for i in range(3)]
    try:
        rng = np.random.default_rng(2026)
        for i in range(160):
            name = list(fz.CASES)[i % len(fz.CASES)]
            fz.CASES[name](np.random.default_rng(int(rng.integers(0, 1 << 31))), k)
    finally:
        for i, v in enumerate(saved):
            ctx.set_tuning(i, v)
