"""Constant-kernel generators (host, NumPy) against the golden vectors produced by the reference's own
generators (tests/golden/make_golden.py) and against the literals of the reference's unit tests."""
import math

import numpy as np
import numpy.testing as npt
import pytest

from pysilent_amd import constant_convolutions as cc
from pysilent_amd.util.attractor import euclidian_attractor_function_generator, linear_attractor_function_generator
from pysilent_amd.util.normalize import normalize_tensor_positive_negative
from pysilent_amd.util.orientation import above_axis_simplex_coordinates, simplex_coordinates

TOL = 1e-12


def test_center_surround_1d_literal():
    # literal of the reference's tests/test_center_surround_tensors.py:8-22
    got = cc.center_surround_tensor(1, [0, 1, 0], [1, 0, 0], [0, 0, 1], [1, 0, 0])
    want = np.zeros((3, 3, 3))
    want[0, 2, 0] = 1.0
    want[1, 1, 0] = 2.0
    want[2, 2, 0] = 1.0
    npt.assert_array_equal(got, want)


def test_center_surround_2d_literal():
    # reference tests/test_center_surround_tensors.py:24-63: surround taps 1/sqrt(manhattan), centre = their sum
    got = cc.center_surround_tensor(2, [0, 1, 0], [1, 0, 0], [0, 0, 1], [1, 0, 0])
    want = np.zeros((3, 3, 3, 3))
    for y in range(3):
        for x in range(3):
            d = abs(y - 1) + abs(x - 1)
            if d:
                want[y, x, 2, 0] = 1.0 / math.sqrt(d)
    want[1, 1, 1, 0] = 6.82842712
    npt.assert_array_almost_equal(got, want, decimal=6)


def test_normalize_literals_and_in_place():
    # reference tests/test_normalize_center_surround.py:9-26
    t1 = np.squeeze(cc.center_surround_tensor(1, [1], [1], [1], [-1]))
    npt.assert_array_almost_equal(t1, [-1, 2, -1])
    npt.assert_array_almost_equal(normalize_tensor_positive_negative(t1), [-.5, 1, -.5])
    t2 = np.squeeze(cc.center_surround_tensor(2, [1], [1], [1], [-1]))
    npt.assert_array_almost_equal(t2, [[-0.70710678, -1., -0.70710678], [-1., 6.82842712, -1.],
                                       [-0.70710678, -1., -0.70710678]])
    want = [[-0.10355339, -0.14644661, -0.10355339], [-0.14644661, 1., -0.14644661],
            [-0.10355339, -0.14644661, -0.10355339]]
    out = normalize_tensor_positive_negative(t2)
    npt.assert_array_almost_equal(out, want)
    npt.assert_array_almost_equal(t2, want)        # in place
    assert out is t2


def test_simplex_literals():
    # reference tests/test_simplex_coordinates.py:9-22
    npt.assert_array_almost_equal(simplex_coordinates(2), [[1., 0.], [-0.5, 0.8660254], [-0.5, -0.8660254]])
    npt.assert_array_almost_equal(simplex_coordinates(3), [[1., 0., 0.], [-0.33333333, 0.94280904, 0.],
                                                           [-0.33333333, -0.47140452, 0.81649658],
                                                           [-0.33333333, -0.47140452, -0.81649658]])


def test_generation_time_bound():
    # the reference's only perf assertion (tests/test_center_surround_tensors.py:66-74): <= 1 s per n
    import time
    for n in range(1, 11):
        t = time.time()
        cc.center_surround_tensor(n, [0, 1, 0], [1, 0, 0], [0, 0, 1], [1, 0, 0])
        assert time.time() - t <= 1.0, n


CASES = {
    "simplex_coordinates_2": lambda g: simplex_coordinates(2),
    "simplex_coordinates_3": lambda g: simplex_coordinates(3),
    "above_axis_simplex_coordinates_2": lambda g: above_axis_simplex_coordinates(2),
    "center_surround_1d_test": lambda g: cc.center_surround_tensor(1, [0, 1, 0], [1, 0, 0], [0, 0, 1], [1, 0, 0]),
    "center_surround_2d_test": lambda g: cc.center_surround_tensor(2, [0, 1, 0], [1, 0, 0], [0, 0, 1], [1, 0, 0]),
    "center_surround_3d_test": lambda g: cc.center_surround_tensor(3, [0, 1, 0], [1, 0, 0], [0, 0, 1], [1, 0, 0]),
    "cs_gray_raw": lambda g: cc.center_surround_tensor(2, [1], [1], [1], [-1]),
    "cs_gray_norm": lambda g: normalize_tensor_positive_negative(cc.center_surround_tensor(2, [1], [1], [1], [-1])),
    "cs_gray_1d_norm": lambda g: normalize_tensor_positive_negative(cc.center_surround_tensor(1, [1], [1], [1], [-1])),
    "normalize_4_2_out": lambda g: normalize_tensor_positive_negative(g["normalize_4_2_in"].copy(), 4.0, 2.0),
    "midget_rgc_2": lambda g: cc.midget_rgc(2),
    "rgby_2": lambda g: cc.rgby(2),
    "rgby_3_2": lambda g: cc.rgby_3(2),
    "rgb_2d_stripe_tensors": lambda g: cc.rgb_2d_stripe_tensors(),
    "rgb_2d_end_tensors": lambda g: cc.rgb_2d_end_tensors(),
    "blur_tensor_2_7": lambda g: cc.blur_tensor(2, 7),
    "blur_tensor_2_3": lambda g: cc.blur_tensor(2, 3),
    "blur_tensor_2_5_1_1": lambda g: cc.blur_tensor(2, 5, 1, 1),
    "end_bank_gray_4": lambda g: np.moveaxis(cc.end_bank(4), 3, 0)[..., None],
    "end_bank_gray_8": lambda g: np.moveaxis(cc.end_bank(8), 3, 0)[..., None],
    "end_tensor_v30_rgb": lambda g: cc.end_tensor(np.array([3., 0.]), [1, 0, 0], [.25, .125, .125], [1, 0, 0],
                                                  [.5, -.5, -.5]),
    "simplex_end_tensors_rgb": lambda g: np.stack(cc.simplex_end_tensors(
        2, [[1, 0, 0], [0, 1, 0], [0, 0, 1]], [[.25, .125, .125], [.125, .25, .125], [.125, .125, .25]],
        [[1, 0, 0], [0, 1, 0], [0, 0, 1]], [[.5, -.5, -.5], [-.5, .5, -.5], [-.5, -.5, .5]]), 0),
    "stripe_tensor_x_gray": lambda g: cc.stripe_tensor([1., 0.], [1], [1], [1], [-1]),
    "stripe_tensor_diag_rgb": lambda g: cc.stripe_tensor([0.5, 0.8660254037844386], [1, 1, 1], [4, -1, -1],
                                                         [1, 1, 1], [-4, 1, 1]),
    # 7x7 thick-edge family
    "edge_tensor_x_gray": lambda g: cc.edge_tensor([1., 0.], [1], [1], [1], [-1]),
    "edge_tensor_diag_rgb": lambda g: cc.edge_tensor([-0.5, 0.8660254037844386], [1, 1, 1], [4, -1, -1],
                                                     [1, 1, 1], [-4, 1, 1]),
    "rgb_2d_edge_tensors": lambda g: cc.rgb_2d_edge_tensors(),
    "rgb_2d_edge_tensors_time_diff": lambda g: cc.rgb_2d_edge_tensors_time_diff(),
    "rgb_2d_end_tensors_7x7": lambda g: cc.edge_orientation_detector.rgb_2d_end_tensors(),
    "simplex_edge_tensors_flip0": lambda g: np.stack(cc.simplex_edge_tensors(
        2, [[1, 0, 0]] * 3, [[1, 0, 0], [0, 1, 0], [0, 0, 1]], [[0, 1, 0]] * 3, [[0, 0, 1]] * 3, flip=0), 0),
    "attractor_euclid_n2_p0_nm1": lambda g: euclidian_attractor_function_generator(
        2, max_positive=0.0, max_negative=-1.0)(g["attractor_x"]),
    "attractor_euclid_n2": lambda g: euclidian_attractor_function_generator(2)(g["attractor_x"]),
    "attractor_euclid_n2_neg0": lambda g: euclidian_attractor_function_generator(2, max_negative=0)(g["attractor_x"]),
    "attractor_euclid_n3_p2_n05": lambda g: euclidian_attractor_function_generator(3, 2.0, 0.5)(g["attractor_x"]),
    "attractor_linear": lambda g: linear_attractor_function_generator()(g["attractor_x"]),
    "attractor_linear_p2_n05": lambda g: linear_attractor_function_generator(2.0, 0.5)(g["attractor_x"]),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_generator_matches_reference_golden(golden, name):
    got = np.asarray(CASES[name](golden), dtype=np.float64)
    assert got.shape == golden[name].shape
    npt.assert_allclose(got, golden[name], rtol=0, atol=TOL)


def test_every_golden_is_covered(golden):
    inputs = {"normalize_4_2_in", "attractor_x", "cs_gray_1d_raw"}
    assert set(golden) - inputs <= set(CASES)


def test_scalar_attractors_return_floats():
    assert isinstance(euclidian_attractor_function_generator(2)(0.5), float)
    assert linear_attractor_function_generator()(-0.25) == 0.5


def test_kernel_structure_facts(golden):
    # facts the HIP kernels may rely on (hard part 2)
    rgc = cc.midget_rgc(2)
    off = rgc.copy()
    for c in range(3):
        off[:, :, c, c] = 0
    assert not off.any()                                            # channel-diagonal
    npt.assert_allclose(rgc[rgc > 0].sum(), 4.0)
    npt.assert_allclose(rgc[rgc < 0].sum(), -2.0)
    stripe = cc.rgb_2d_stripe_tensors()
    assert np.array_equal(stripe[:, :, 0], stripe[:, :, 1]) and np.array_equal(stripe[:, :, 0], stripe[:, :, 2])
    blur = cc.blur_tensor(2, 7)
    assert (blur == blur[:, :, :1, :1]).all() and blur[3, 3, 0, 0] == 1.0 and blur[3, 0, 0, 0] == 1.0 / 7
    with pytest.raises(ValueError):
        cc.stripe_tensor([1., 0.], [1, 1, 1], [1, 0], [1, 1, 1], [1, 0])   # non-square channel lists


def test_edge_tensor_facet_passes_through_tap_1_not_the_centre():
    # reference behaviour kept on purpose (edge_tensor.py:56): distance is measured from tap (1, 1) of the 7-grid
    z = cc.edge_tensor([1., 0.], [1], [1], [1], [-1])[:, :, 0, 0]
    assert (z[1] == 0).all() and (z[0] < 0).all() and (z[2:] > 0).all()
    npt.assert_allclose(z[0].sum(), -1.0, atol=1e-12)          # normalised: sum- = -1, sum+ = +1
    npt.assert_allclose(z[2:].sum(), 1.0, atol=1e-12)
    with pytest.raises(ValueError):
        cc.edge_tensor([1., 0.], [1, 1], [1], [1, 1], [1])      # non-square channel lists, like the reference
