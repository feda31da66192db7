"""``PyramidDisplayer``: base class of the reference's camera apps, slam_recognition/pyramid_displayer.py:18-61.

Kept: the constructor arguments and the ``callback`` / ``display`` contract (``display`` = callback output / 255,
pyramid_displayer.py:35-40).  Not kept: ``run_camera`` (cvpubsubs / OpenCV capture and windows, SURVEY.md
section 8: out of scope) -- feed frames to ``callback`` / ``display`` from whatever capture loop the application has.
"""
import math as m

import numpy as np


class PyramidDisplayer(object):
    def __init__(self, output_size=(int(36 * 8), int(24 * 8)), output_colors=3, zoom_ratio=m.e ** .5):
        """Generates several smaller images at different zoom levels from one input image."""
        self.output_size = output_size
        self.output_colors = output_colors
        self.zoom_ratio = zoom_ratio

    def callback(self, frame, cam_id=None):
        return [frame]

    def display(self, frame, cam_id=None):
        frame_from_callback = self.callback(frame, cam_id)
        return [np.array(frame_from_callback[x]) / 255.0 for x in range(len(frame_from_callback))]

    def run_camera(self, *args, **kwargs):
        raise NotImplementedError("camera capture and windows (cvpubsubs) are outside this build: call "
                                  "display(frame, cam_id) from your own capture loop")
