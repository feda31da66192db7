"""pysilent_amd -- MI355X (gfx950) implementation of pySILEnT's scale-space center-surround /
oriented line-end detector hot path, behind the reference's own Python filter API.

Layout mirrors ``slam_recognition`` for the path in scope:
    pysilent_amd.constant_convolutions   kernel generators (host, NumPy float64)
    pysilent_amd.filters                 rgc_filter, rgby_filter, orientation_filter
    pysilent_amd.util                    apply_filter, regulator, selection, color, energy, zoom, ...
    pysilent_amd.pipeline                LineEndPipeline: the whole pass, batched, device-resident
    pysilent_amd.distributed             frame sharding over ranks + one RCCL broadcast of the constants
The compute lives in ``lib/libsilent_hip.so`` (hand-written HIP, C ABI in include/silent_hip.h).
There is NO CPU fallback: without the library and a gfx950 GPU the filters raise.
"""
__version__ = "0.1.0"

from .constant_convolutions.center_surround import center_surround_tensor
from .constant_convolutions.edge_orientation_detector import stripe_tensor, simplex_stripe_tensors
from .util import zoom
from ._runtime import PackedPyramid, PyramidPlan, get_context, device_count
