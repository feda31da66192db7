"""Constant (hand-constructed, never trained) convolution kernels -- host side, NumPy float64.

Same public names as ``slam_recognition/constant_convolutions/__init__.py:1-5``.
"""
from .center_surround import center_surround_tensor, midget_rgc, midget_rgc_full, rgby, rgby_3
from . import edge_orientation_detector
from .edge_orientation_detector import stripe_tensor, simplex_stripe_tensors, rgb_2d_stripe_tensors
from .edge_orientation_detector import (edge_tensor, simplex_edge_tensors, rgb_2d_edge_tensors,
                                        rgb_2d_edge_tensors_time_diff)
from .gaussian_blur import blur_tensor, blur_profile
from .oriented_end_detector import end_tensor, simplex_end_tensors, rgb_2d_end_tensors, end_bank
