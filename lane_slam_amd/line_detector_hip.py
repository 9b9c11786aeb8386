"""Drop-in LineDetectorInterface plugin backed by the HIP library.

Mirrors /root/reference/src/line_detector/include/line_detector/line_detector_lsd.py:11-142
(`LineDetectorLSD`): same constructor contract (`configuration` dict with exactly the 13
keys of :20-34, ValueError on extra/missing keys like duckietown_utils/parameters.py:15-23),
same `setImage(bgr)` / `detectLines(color)` / `getImage()` methods, same return types:
`Detections(lines, normals, area, centers)` with `lines` float32 (N,4) -- or an empty list
when nothing is found (:68-71) -- `normals` float64 (N,2), `area` the dilated colour mask,
`centers` float32 (N,2).  In the node's yaml:

    detector:
      - lane_slam_amd.LineDetectorHIP
      - configuration: { ...same 13 keys... }
"""
import copy
import ctypes
from collections import namedtuple

import numpy as np

from . import _lib
from .config import DETECTOR_KEYS, default_config
from .frontend import FrontEnd

Detections = namedtuple("Detections", ["lines", "normals", "area", "centers"])   # line_detector_interface.py:6-7

_COLOR_CODE = {"white": 0, "yellow": 1, "red": 2}


class LineDetectorInterface(object):
    """line_detector_interface.py:10-19"""

    def setImage(self, bgr):
        raise NotImplementedError

    def detectLines(self, color):
        """ Returns a tuple of class Detections """
        raise NotImplementedError


class LineDetectorHIP(LineDetectorInterface):
    def __init__(self, configuration, device=0, max_lines_per_color=2048, lsd_seed_order="opencv32"):
        """configuration: the reference's 13 keys (line_detector_lsd.py:20-34), nothing else.  lsd_seed_order (keyword, not a
        configuration key): "opencv30" or "opencv32" -- which OpenCV's LSD seed order inside a gradient bin (lf_config.lsd_seed_order;
        ROS Kinetic's 3.3.1 is "opencv32")."""
        if not isinstance(configuration, dict):
            raise ValueError("Expecting a dict, obtained %r" % (configuration,))
        configuration = copy.deepcopy(configuration)
        extra = set(configuration) - set(DETECTOR_KEYS)
        missing = set(DETECTOR_KEYS) - set(configuration)
        if extra or missing:
            raise ValueError("Error while loading configuration for %r from %r.\nExtra parameters: %r\n"
                             "Missing parameters: %r\n" % (self, configuration, extra, missing))
        for k in DETECTOR_KEYS:
            v = configuration[k]
            if isinstance(v, list) and len(v) == 3:
                v = np.array(v)
            setattr(self, k, v)
        self._configuration = configuration
        self._seed_order = lsd_seed_order
        self._device = device
        self._cap = int(max_lines_per_color)
        self._fe = None
        self._shape = None
        self._bufs = None
        self.bgr = np.empty(0)
        _lib.load()          # fail at construction time if the HIP library is missing

    def _frontend(self, rows, cols):
        if self._fe is None or self._shape != (rows, cols):
            if self._fe is not None:
                self._fe.close()
            cfg = default_config("parity")
            cfg["in_size"] = [rows, cols]
            cfg["img_size"] = [rows, cols]
            cfg["top_cutoff"] = 0
            cfg["detector"] = {k: (list(map(int, v)) if hasattr(v, "__len__") else v)
                               for k, v in self._configuration.items()}
            cfg["lsd"]["seed_order"] = self._seed_order
            self._fe = FrontEnd(cfg, device=self._device, max_frames=1, max_lines_per_color=self._cap)
            self._shape = (rows, cols)
        return self._fe

    def setImage(self, bgr):
        bgr = np.asarray(bgr)
        if bgr.ndim != 3 or bgr.shape[2] != 3 or bgr.dtype != np.uint8:
            raise ValueError("setImage expects a uint8 HxWx3 BGR image")
        self.bgr = np.copy(bgr)                                   # line_detector_lsd.py:136
        fe = self._frontend(bgr.shape[0], bgr.shape[1])
        img = np.ascontiguousarray(self.bgr)
        fe._check(fe.lib.lf_set_image(fe.h, img.ctypes.data_as(ctypes.c_void_p), img.shape[0], img.shape[1],
                                      img.strides[0]))

    def detectLines(self, color):
        if color not in _COLOR_CODE:
            raise Exception("Error: Undefined color strings...")   # line_detector_lsd.py:48-49
        if self._fe is None:
            raise Exception("detectLines called before setImage")
        fe = self._fe
        cap = self._cap
        # the receiving arrays are made once per detector (the library fills them from pinned memory it fetched with the
        # image: no device work here); what is returned are copies of the used part, as the reference returns fresh arrays
        if self._bufs is None or self._bufs[0].shape[0] != cap:
            self._bufs = (np.empty((cap, 4), np.float32), np.empty((cap, 2), np.float64), np.empty((cap, 2), np.float32))
            self._bufp = tuple(b.ctypes.data_as(ctypes.c_void_p) for b in self._bufs)
        lines, normals, centers = self._bufs
        area = np.empty(self._shape, np.uint8)
        n = ctypes.c_int()
        fe._check(fe.lib.lf_detect_lines(fe.h, _COLOR_CODE[color], self._bufp[0], self._bufp[1], self._bufp[2],
                                         area.ctypes.data_as(ctypes.c_void_p), cap, ctypes.byref(n)))
        k = n.value
        if k == 0:
            return Detections(lines=[], normals=[], area=area, centers=[])   # line_detector_lsd.py:68-71,87-88
        return Detections(lines=lines[:k].copy(), normals=normals[:k].copy(), area=area, centers=centers[:k].copy())

    def getImage(self):
        return self.bgr


class LineDetectorEDLines(LineDetectorHIP):
    """The same plugin interface over the EDLines detector of the reference's line_descriptor library
    (ref: src/line_descriptor/src/binary_descriptor_custom.cpp:1374-2751, BinaryDescriptor::detect) instead of OpenCV's
    LSD -- SURVEY 8f-4's "alternative detector plugin".  Same constructor contract and the same 13 configuration keys
    as LineDetectorLSD (the HSV boxes and dilation_kernel_size decide the colour of a line; canny_thresholds is
    accepted and unused: EDLines works on the gray image, not on Canny edges).  Contract (this package's own, stated in
    include/lanefront.h at lf_set_image_edlines): a detected line belongs to every colour whose dilated mask covers its
    centre; normals, centres and endpoint ordering as in LineDetectorLSD._findNormal / _correctPixelOrdering.

        detector:
          - lane_slam_amd.LineDetectorEDLines
          - configuration: { ...same 13 keys... }

    Optional keyword `edlines`: dict overriding EDLineDetector's defaults (gradient_threshold 80, anchor_threshold 8,
    scan_intervals 2, min_line_len 15, line_fit_err_threshold 1.6)."""

    def __init__(self, configuration, device=0, max_lines_per_color=2048, edlines=None):
        LineDetectorHIP.__init__(self, configuration, device=device, max_lines_per_color=max_lines_per_color)
        self._edlines = dict(edlines or {})

    def setImage(self, bgr):
        bgr = np.asarray(bgr)
        if bgr.ndim != 3 or bgr.shape[2] != 3 or bgr.dtype != np.uint8:
            raise ValueError("setImage expects a uint8 HxWx3 BGR image")
        self.bgr = np.copy(bgr)
        fe = self._frontend(bgr.shape[0], bgr.shape[1])
        img = np.ascontiguousarray(self.bgr)
        p = fe.edlines_params(**self._edlines)
        fe._check(fe.lib.lf_set_image_edlines(fe.h, img.ctypes.data_as(ctypes.c_void_p), img.shape[0], img.shape[1], img.strides[0],
                                              ctypes.byref(p)))
