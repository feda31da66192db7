"""Recovery term of the boosting update.  Mirror of slam_recognition/util/energy/recovery.py:4-22: the reference
builds the term as graph nodes; here ``generate_recovery`` only selects the mode that silent_boosting_step
evaluates on the GPU (same argument meaning, same ValueError)."""
from ... import _lib

RECOVERY_AMOUNT = 10.0        # generate_constant_recovery, recovery.py:4
RECOVERY_PERCENTAGE = 0.8     # generate_input_based_recovery, recovery.py:8


def recovery_mode(is_input_based=False, is_constant=True):
    if not is_input_based and not is_constant:
        raise ValueError("You must choose a type of recovery")
    return (_lib.RECOVERY_CONSTANT if is_constant else 0) | (_lib.RECOVERY_INPUT if is_input_based else 0)


def generate_constant_recovery(tensor_in, recovery_amount=RECOVERY_AMOUNT):
    """recovery.py:4-5: ones_like(tensor_in) * recovery_amount -- a constant map (no arithmetic on the input, so a NaN
    input does not show): a filled buffer of the input's kind (NumPy array, torch tensor, or PackedPyramid)."""
    import numpy as np
    from ... import _runtime
    if isinstance(tensor_in, _runtime.PackedPyramid):
        data = generate_constant_recovery(tensor_in.data, recovery_amount)
        return _runtime.PackedPyramid(data, tensor_in.extents, tensor_in.channels, tensor_in.n_frames)
    if _runtime.is_torch_tensor(tensor_in):
        import torch
        return torch.full_like(tensor_in, float(recovery_amount), dtype=torch.float32)      # memory initialisation only
    return np.full(np.shape(tensor_in), np.float32(recovery_amount), np.float32)


def generate_input_based_recovery(tensor_in, recovery_percentage=RECOVERY_PERCENTAGE):
    """recovery.py:8-9: tensor_in * recovery_percentage (silent_affine_clip on the GPU)."""
    from ... import _runtime
    return _runtime.affine_clip(tensor_in, mul=recovery_percentage)


def generate_recovery(tensor_in, is_input_based=False, is_constant=True):
    """Mirror of slam_recognition/util/energy/recovery.py:12-22 under its own name: the recovery MAP of the boosting
    update, evaluated eagerly on the GPU (the reference returns the symbolic node).  ``get_boosting`` does not call
    this -- silent_boosting_step evaluates the same term inside its update kernel -- it exists so that code written
    against the reference's module finds the function.  Same ValueError for "neither"."""
    from ... import _runtime
    recovery_mode(is_input_based, is_constant)           # raises ValueError("You must choose a type of recovery")
    if is_input_based and not is_constant:
        return generate_input_based_recovery(tensor_in)
    if is_constant and not is_input_based:
        return generate_constant_recovery(tensor_in)
    # tf.maximum(input * 0.8, 10): one affine + lower clip (Eigen's (x < lo) ? lo : x keeps a NaN a NaN)
    return _runtime.affine_clip(tensor_in, mul=RECOVERY_PERCENTAGE, lo=RECOVERY_AMOUNT)
