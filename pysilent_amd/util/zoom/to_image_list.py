"""Pyramid tensor -> list of uint8 images.  Mirror of slam_recognition/util/zoom/to_image_list.py:7-15
(host-side convenience for display; pure NumPy by nature)."""
import numpy as np


def zoom_tensor_to_image_list(zoom, axis=2):
    zoom = np.asarray(zoom)
    return [np.squeeze(zoom[p:p + 1]).astype(np.uint8) for p in range(zoom.shape[0])]
