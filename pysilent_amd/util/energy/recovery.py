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
