"""Utility layer; same sub-module names as slam_recognition/util/__init__.py:1-2 (minus the out-of-scope
math / relativity / index_tensor / centroids, SURVEY.md sections 2 and 8f)."""
from . import attractor, color, energy, normalize, orientation, regulator, selection, zoom
from . import apply_filter, get_dimensions
