"""lanefront: MI355X-native line-feature front end for lane-slam (detect, describe,
ground-project, sanity-filter, associate) behind a C ABI (include/lanefront.h).

The package is a thin host-side mirror of the reference's interfaces for this path:
  LineDetectorHIP  <->  line_detector.LineDetectorLSD (LineDetectorInterface plugin)
  LineDetectorEDLines   the same plugin interface over the line_descriptor library's EDLines detector; FrontEnd.keylines_batch /
                        describe_keylines <-> BinaryDescriptor::detect / compute with octaves
  LineAssociator   <->  line_associator node (a stub in the reference) + show_map's segment store: device-resident
                        live map, MFMA Hamming association, colour gating, append / merge updates (lf_map_*)
  BinaryDescriptorMatcher <-> the matcher's dataset form (add / train / match / knnMatch / radiusMatch over several train images, imgIdx)
  FrontEnd         <->  batch form of line_detector_node / ground_projection_node /
                        line_sanity_node callbacks + BinaryDescriptor / BinaryDescriptorMatcher
There is no CPU fallback: importing works anywhere, but creating a detector without the
HIP library or without a GPU raises.
"""
from .config import (COLOR_NAMES, DEFAULT_DETECTOR_CONFIGURATION, RED, WHITE, YELLOW, default_config)
from .frontend import FrontEnd, LanefrontError, Segments
from .line_associator import LineAssociator
from .matcher import BinaryDescriptorMatcher, BinaryDescriptorParams, DMatch
from .line_detector_hip import Detections, LineDetectorEDLines, LineDetectorHIP, LineDetectorInterface

__all__ = ["BinaryDescriptorMatcher", "BinaryDescriptorParams", "DMatch", "LineAssociator", "FrontEnd", "LanefrontError", "Segments", "LineDetectorHIP", "LineDetectorEDLines", "LineDetectorInterface", "Detections",
           "default_config", "DEFAULT_DETECTOR_CONFIGURATION", "WHITE", "YELLOW", "RED", "COLOR_NAMES"]
