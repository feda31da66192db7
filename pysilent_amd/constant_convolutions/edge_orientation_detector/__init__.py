"""Oriented edge kernels: 3x3 stripes (thin edges) and the 7x7 thick-edge family.  Same names as
slam_recognition/constant_convolutions/edge_orientation_detector/__init__.py:1-2."""
from .stripe_tensor import stripe_tensor, simplex_stripe_tensors, rgb_2d_stripe_tensors
from .edge_tensor import (edge_tensor, simplex_edge_tensors, rgb_2d_edge_tensors, rgb_2d_end_tensors,
                          rgb_2d_edge_tensors_time_diff)
