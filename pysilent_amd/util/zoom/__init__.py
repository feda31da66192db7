"""The zoom pyramid.  ``from_image`` / ``to_image_list`` mirror slam_recognition/util/zoom/__init__.py:1-2."""
from .from_image import image_to_zoom_tensor as from_image, classic_pyramid, classic_levels, reference_levels
from .to_image_list import zoom_tensor_to_image_list as to_image_list
