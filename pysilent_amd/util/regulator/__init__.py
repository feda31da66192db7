"""``regulate_tensor``: blur-based divisive normalisation ("per-level normalization" of the north star).

Drop-in for slam_recognition/util/regulator/gaussian_regulator_tensor.py:10-36:
    y = x * (regulation_value / pow(min(conv2d(x, blur), 1), regulation_root))
computed by ONE gfx950 kernel (blur stencil + pointwise epilogue; the blurred map never reaches HBM).
``flat_policy``: "ieee" replicates the reference literally (0 * inf = NaN where a whole window is 0),
"zero" returns 0 there (hard part 3).
"""
from ... import _runtime
from ..get_dimensions import get_dimensions


def regulate_tensor(input_tensor, blur_tensor, regulation_value, regulation_root=1.0 / 2.0, strides=(1, 1, 1, 1),
                    padding='SAME', flat_policy="ieee"):
    get_dimensions(input_tensor)
    if tuple(strides) != (1, 1, 1, 1) or padding != 'SAME':
        raise ValueError("regulate_tensor: only strides=(1,1,1,1), padding='SAME' (all the reference uses)")
    return _runtime.regulate(input_tensor, blur_tensor, regulation_value, regulation_root, flat_policy)
