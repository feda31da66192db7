"""Utility layer; same sub-module names as slam_recognition/util/__init__.py:1-2 (minus the out-of-scope
math / relativity)."""
from . import attractor, color, energy, normalize, orientation, regulator, selection, zoom
from . import apply_filter, get_dimensions, index_tensor
from .centroids import additive_filter, get_centroids
