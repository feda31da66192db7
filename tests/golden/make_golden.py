#!/usr/bin/env python3
"""Generate golden vectors for the constant-kernel generators.

Runs ONLY in the build container (needs /root/reference, which never travels to the
GPU box).  It imports the reference's pure-NumPy generator modules for real and
stores their outputs as float64 arrays in ``tests/golden/kernels.npz``.

The reference package imports TensorFlow at module scope (util/color/*.py) although
none of the generators use it, and TensorFlow is not installable here.  An *inert*
``tensorflow`` module (attribute access yields further empty modules; it implements
no TensorFlow behaviour whatsoever) is registered in ``sys.modules`` so that those
imports resolve.  Nothing that would execute TensorFlow code is called.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import math
import os
import sys
import types

import numpy as np

REF = os.environ.get("SILENT_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


class _Inert(types.ModuleType):
    """Module whose every attribute is another inert module (no behaviour)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        full = self.__name__ + "." + name
        mod = sys.modules.get(full)
        if mod is None:
            mod = _Inert(full)
            sys.modules[full] = mod
        if name == "Tensor":
            return type("Tensor", (), {})
        return mod


def _install_inert_tensorflow():
    for name in ("tensorflow", "tensorflow.python", "tensorflow.python.ops",
                 "tensorflow.python.ops.array_ops", "tensorflow.python.ops.math_ops",
                 "tensorflow.python.framework", "tensorflow.python.framework.ops",
                 "tensorflow.python.framework.dtypes"):
        sys.modules.setdefault(name, _Inert(name))


def main():
    sys.dont_write_bytecode = True
    _install_inert_tensorflow()
    sys.path.insert(0, REF)
    from slam_recognition.constant_convolutions.center_surround import (
        center_surround_tensor, midget_rgc, rgby, rgby_3)
    from slam_recognition.constant_convolutions.edge_orientation_detector.stripe_tensor import (
        rgb_2d_stripe_tensors, stripe_tensor)
    import importlib
    ref_edge = importlib.import_module("slam_recognition.constant_convolutions.edge_orientation_detector.edge_tensor")
    from slam_recognition.constant_convolutions.oriented_end_detector import (
        end_tensor, rgb_2d_end_tensors, simplex_end_tensors)
    from slam_recognition.constant_convolutions.gaussian_blur.gaussian_blur import blur_tensor
    from slam_recognition.util.normalize import normalize_tensor_positive_negative
    from slam_recognition.util.orientation import (
        simplex_coordinates, above_axis_simplex_coordinates)
    from slam_recognition.util.attractor import (
        euclidian_attractor_function_generator, linear_attractor_function_generator)

    out = {}
    out["simplex_coordinates_2"] = simplex_coordinates(2)
    out["simplex_coordinates_3"] = simplex_coordinates(3)
    out["above_axis_simplex_coordinates_2"] = above_axis_simplex_coordinates(2)

    out["center_surround_1d_test"] = center_surround_tensor(1, [0, 1, 0], [1, 0, 0], [0, 0, 1], [1, 0, 0])
    out["center_surround_2d_test"] = center_surround_tensor(2, [0, 1, 0], [1, 0, 0], [0, 0, 1], [1, 0, 0])
    out["center_surround_3d_test"] = center_surround_tensor(3, [0, 1, 0], [1, 0, 0], [0, 0, 1], [1, 0, 0])
    cs1 = center_surround_tensor(2, [1], [1], [1], [-1])
    out["cs_gray_raw"] = cs1.copy()
    out["cs_gray_norm"] = normalize_tensor_positive_negative(cs1.copy())
    cs1d = center_surround_tensor(1, [1], [1], [1], [-1])
    out["cs_gray_1d_raw"] = cs1d.copy()
    out["cs_gray_1d_norm"] = normalize_tensor_positive_negative(cs1d.copy())
    out["normalize_4_2_in"] = np.array([[-3.0, 0.5, 2.0], [0.0, -1.0, 4.0]])
    out["normalize_4_2_out"] = normalize_tensor_positive_negative(out["normalize_4_2_in"].copy(), 4.0, 2.0)

    out["midget_rgc_2"] = midget_rgc(2)
    out["rgby_2"] = rgby(2)
    out["rgby_3_2"] = rgby_3(2)
    out["rgb_2d_stripe_tensors"] = rgb_2d_stripe_tensors()
    out["rgb_2d_end_tensors"] = rgb_2d_end_tensors()
    out["blur_tensor_2_7"] = blur_tensor(2, 7)
    out["blur_tensor_2_3"] = blur_tensor(2, 3)
    out["blur_tensor_2_5_1_1"] = blur_tensor(2, 5, 1, 1)

    for K in (4, 8):
        bank = []
        for k in range(K):
            v = 3.0 * np.array([math.cos(k * math.pi / K), math.sin(k * math.pi / K)])
            bank.append(end_tensor(v, [1], [1], [1], [-1]))
        out["end_bank_gray_%d" % K] = np.stack(bank, 0)      # [K,3,3,1,1]
    out["end_tensor_v30_rgb"] = end_tensor(np.array([3.0, 0.0]), [1, 0, 0], [.25, .125, .125],
                                           [1, 0, 0], [.5, -.5, -.5])
    st = simplex_end_tensors(2, [[1, 0, 0], [0, 1, 0], [0, 0, 1]],
                             [[.25, .125, .125], [.125, .25, .125], [.125, .125, .25]],
                             [[1, 0, 0], [0, 1, 0], [0, 0, 1]],
                             [[.5, -.5, -.5], [-.5, .5, -.5], [-.5, -.5, .5]])
    out["simplex_end_tensors_rgb"] = np.stack(st, 0)
    out["stripe_tensor_x_gray"] = stripe_tensor([1.0, 0.0], [1], [1], [1], [-1])
    out["stripe_tensor_diag_rgb"] = stripe_tensor([0.5, 0.8660254037844386], [1, 1, 1], [4, -1, -1],
                                                  [1, 1, 1], [-4, 1, 1])

    # 7x7 thick-edge family
    out["edge_tensor_x_gray"] = ref_edge.edge_tensor([1.0, 0.0], [1], [1], [1], [-1])
    out["edge_tensor_diag_rgb"] = ref_edge.edge_tensor([-0.5, 0.8660254037844386], [1, 1, 1], [4, -1, -1],
                                                       [1, 1, 1], [-4, 1, 1])
    out["rgb_2d_edge_tensors"] = ref_edge.rgb_2d_edge_tensors()
    out["rgb_2d_edge_tensors_time_diff"] = ref_edge.rgb_2d_edge_tensors_time_diff()
    out["rgb_2d_end_tensors_7x7"] = ref_edge.rgb_2d_end_tensors()
    out["simplex_edge_tensors_flip0"] = np.stack(ref_edge.simplex_edge_tensors(
        2, [[1, 0, 0]] * 3, [[1, 0, 0], [0, 1, 0], [0, 0, 1]], [[0, 1, 0]] * 3, [[0, 0, 1]] * 3, flip=0), 0)
    f2n = euclidian_attractor_function_generator(2, max_positive=0.0, max_negative=-1.0)

    xs = np.array([-2.5, -1.0, -0.25, 0.0, 0.25, 0.5, 1.0, 1.4142135623730951, 2.0, 3.0])
    f2 = euclidian_attractor_function_generator(2)
    f2b = euclidian_attractor_function_generator(2, max_negative=0)
    f3 = euclidian_attractor_function_generator(3, 2.0, 0.5)
    fl = linear_attractor_function_generator()
    fl2 = linear_attractor_function_generator(2.0, 0.5)
    out["attractor_x"] = xs
    out["attractor_euclid_n2"] = np.array([f2(x) for x in xs])
    out["attractor_euclid_n2_neg0"] = np.array([f2b(x) for x in xs])
    out["attractor_euclid_n3_p2_n05"] = np.array([f3(x) for x in xs])
    out["attractor_euclid_n2_p0_nm1"] = np.array([f2n(x) for x in xs])
    out["attractor_linear"] = np.array([fl(x) for x in xs])
    out["attractor_linear_p2_n05"] = np.array([fl2(x) for x in xs])

    path = os.path.join(HERE, "kernels.npz")
    np.savez(path, **{k: np.asarray(v, dtype=np.float64) for k, v in out.items()})
    for k, v in sorted(out.items()):
        print("%-32s %s" % (k, np.asarray(v).shape))
    print("wrote", path)


if __name__ == "__main__":
    main()
