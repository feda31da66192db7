"""Stateful boosting / exhaustion.

Mirror of slam_recognition/util/energy/boosting.py:6-42.  The reference keeps the state in a ``tf.Variable`` that
``session.run(update_energy)`` assigns; here the state is a plain float32 tensor of the input's kind (ndarray,
torch GPU tensor or PackedPyramid) that ``get_boosting`` advances IN PLACE, so checkpointing a stream is a copy
of that tensor.  One state per stream: feed the frames of a stream to the same state, in order.
"""
import numpy as np

from ... import _runtime
from ..get_dimensions import get_dimensions
from .recovery import recovery_mode


def initialize_boosting(input_tensor, initial_multiplier=8):
    """ones_like(input) * initial_multiplier (boosting.py:6-7), of the input's kind."""
    get_dimensions(input_tensor)
    if isinstance(input_tensor, _runtime.PackedPyramid):
        data = initialize_boosting_flat(input_tensor.data, initial_multiplier)
        return _runtime.PackedPyramid(data, input_tensor.extents, input_tensor.channels, input_tensor.n_frames)
    return initialize_boosting_flat(input_tensor, initial_multiplier)


def initialize_boosting_flat(t, initial_multiplier):
    if isinstance(t, np.ndarray):
        return np.full(t.shape, initial_multiplier, np.float32)
    import torch
    return torch.full(tuple(t.shape), float(initial_multiplier), dtype=torch.float32, device=t.device)


def get_boosting(input_tensor, exhaustion_tensor, exhaustion_max=1, excitation_max=1, input_based_recovery=False,
                 constant_recovery=True, for_visualizing=False):
    """(has_fired, update_energy) like boosting.py:10-42; ``exhaustion_tensor`` holds the new state afterwards."""
    get_dimensions(input_tensor)
    mode = recovery_mode(input_based_recovery, constant_recovery)
    return _runtime.boosting_step(input_tensor, exhaustion_tensor, exhaustion_max, excitation_max, mode,
                                  bool(for_visualizing))
