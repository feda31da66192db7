"""The public filter API -- same names and positional arguments as slam_recognition/filters/__init__.py:1-3.

Each filter is ``relu(conv2d_SAME(tensor, <constant kernel>))`` and runs as one gfx950 stencil launch;
``orientation_filter`` adds the divisive regulator.  Unlike the reference (which returns symbolic
tf.Tensors evaluated later by session.run) these are eager and return float32 data of the input's kind
(np.ndarray -> np.ndarray, torch GPU tensor -> torch GPU tensor, PackedPyramid -> PackedPyramid).
"""
from .orientation import orientation_filter
from .rgby import rgby_filter
from .rgc import rgc_filter
