"""BinaryDescriptorMatcher's dataset form on the device: add / train / clear / match / knnMatch / radiusMatch over the
descriptors of SEVERAL train images (ref: src/line_descriptor/src/binary_descriptor_matcher.cpp:70-111, 117-195, 339-425,
508-595).  Same names and argument meaning as the reference's class; a DMatch is (queryIdx, trainIdx, imgIdx, distance) with
trainIdx the row number in the whole set, as the reference returns it.  The pair forms (query against ONE train matrix) are
FrontEnd.associate / knn_match / radius_match."""
import collections
import ctypes

import numpy as np

from . import _lib

DMatch = collections.namedtuple("DMatch", "queryIdx trainIdx imgIdx distance")


class _LfDmatch(ctypes.Structure):
    _fields_ = [("queryIdx", ctypes.c_int32), ("trainIdx", ctypes.c_int32), ("imgIdx", ctypes.c_int32), ("distance", ctypes.c_float)]


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class BinaryDescriptorParams(object):
    """BinaryDescriptor::Params (ref: src/line_descriptor/src/binary_descriptor_custom.cpp:108-116, 189-204): the four fields with the
    reference's defaults, read() from / write() to a mapping with the keys its cv::FileStorage node uses -- read takes numOfOctave_,
    widthOfBand_ and reductionRatio (not ksize_, exactly as the reference), write adds numOfBand_ = 9.  apply(frontend) hands them to a
    handle (FrontEnd.set_descriptor_params)."""
    NUM_OF_BANDS = 9

    def __init__(self):
        self.numOfOctave_ = 1
        self.widthOfBand_ = 7
        self.reductionRatio = 2
        self.ksize_ = 5

    def read(self, fn):
        # a missing key reads as 0 from a cv::FileNode (the conversion operator of an empty node)
        self.numOfOctave_ = int(fn.get("numOfOctave_", 0))
        self.widthOfBand_ = int(fn.get("widthOfBand_", 0))
        self.reductionRatio = int(fn.get("reductionRatio", 0))
        return self

    def write(self, fs=None):
        fs = {} if fs is None else fs
        fs["numOfOctave_"] = self.numOfOctave_
        fs["numOfBand_"] = self.NUM_OF_BANDS
        fs["widthOfBand_"] = self.widthOfBand_
        fs["reductionRatio"] = self.reductionRatio
        return fs

    def apply(self, frontend):
        return frontend.set_descriptor_params(num_of_octave=self.numOfOctave_, width_of_band=self.widthOfBand_,
                                              reduction_ratio=self.reductionRatio, ksize=self.ksize_)


class BinaryDescriptorMatcher(object):
    def __init__(self, frontend):
        """frontend: the FrontEnd whose handle (device, stream, tie rule) the searches run on.  The dataset lives IN that handle
        (lf_matcher_add / lf_matcher_clear): one matcher per FrontEnd at a time -- a second one would share, and on construction wipe, the
        first one's set, unlike the reference's independent objects -- so a second construction while the first is alive is refused;
        close() (or dropping the object) releases the handle for another."""
        import weakref
        prev = getattr(frontend, "_matcher_ref", None)
        if prev is not None and prev() is not None and not prev()._closed:
            raise ValueError("this FrontEnd already carries a BinaryDescriptorMatcher's dataset: close() it first, or use another FrontEnd")
        self.fe = frontend
        self.lib = frontend.lib
        self._closed = False
        self.fe._check(self.lib.lf_matcher_clear(self.fe.h))
        frontend._matcher_ref = weakref.ref(self)

    def close(self):
        """Empty the set and give the FrontEnd's dataset slot back."""
        if not self._closed:
            self._closed = True
            if getattr(self.fe, "h", None):
                self.lib.lf_matcher_clear(self.fe.h)

    # ---- the set
    def add(self, descriptors):
        """descriptors: a list of [n_i, 32] uint8 matrices, one per train image (:70-80)."""
        for d in descriptors:
            d = np.ascontiguousarray(d, dtype=np.uint8).reshape(-1, 32)
            self.fe._check(self.lib.lf_matcher_add(self.fe.h, _ptr(d), d.shape[0], 0))

    def train(self):
        """(:83-93) nothing to do: the searches always run on everything added so far."""

    def clear(self):
        self.fe._check(self.lib.lf_matcher_clear(self.fe.h))

    def size(self):
        ni, nd = ctypes.c_int(), ctypes.c_int()
        self.fe._check(self.lib.lf_matcher_size(self.fe.h, ctypes.byref(ni), ctypes.byref(nd)))
        return ni.value, nd.value

    def _masks(self, masks, nq):
        if masks is None or len(masks) == 0:
            return None, None
        ni, _ = self.size()
        if len(masks) != ni:
            raise ValueError("the number of images in dataset is %d but %d masks were given" % (ni, len(masks)))
        keep = [None if m is None else np.ascontiguousarray(m, dtype=np.uint8).reshape(-1) for m in masks]
        for m in keep:
            if m is not None and m.shape[0] != nq:
                raise ValueError("a mask must have one byte per query")
        arr = (ctypes.c_void_p * ni)(*[None if m is None else m.ctypes.data for m in keep])
        return arr, keep

    @staticmethod
    def _list(out, a, b):
        return [DMatch(out[i].queryIdx, out[i].trainIdx, out[i].imgIdx, out[i].distance) for i in range(a, b)]

    # ---- the searches
    def match(self, queryDescriptors, masks=None):
        q = np.ascontiguousarray(queryDescriptors, dtype=np.uint8).reshape(-1, 32)
        arr, keep = self._masks(masks, q.shape[0])
        out = (_LfDmatch * max(1, q.shape[0]))()
        n = ctypes.c_int()
        self.fe._check(self.lib.lf_matcher_match(self.fe.h, _ptr(q), q.shape[0], arr, out, ctypes.byref(n)))
        return self._list(out, 0, n.value)

    def knnMatch(self, queryDescriptors, k, masks=None, compactResult=False):
        q = np.ascontiguousarray(queryDescriptors, dtype=np.uint8).reshape(-1, 32)
        arr, keep = self._masks(masks, q.shape[0])
        out = (_LfDmatch * max(1, q.shape[0] * int(k)))()
        off = np.zeros(q.shape[0] + 1, np.int32)
        nl = ctypes.c_int()
        self.fe._check(self.lib.lf_matcher_knn_match(self.fe.h, _ptr(q), q.shape[0], int(k), arr, int(bool(compactResult)), _ptr(off), out, ctypes.byref(nl)))
        return [self._list(out, int(off[i]), int(off[i + 1])) for i in range(nl.value)]

    def radiusMatch(self, queryDescriptors, maxDistance, masks=None, compactResult=False):
        q = np.ascontiguousarray(queryDescriptors, dtype=np.uint8).reshape(-1, 32)
        arr, keep = self._masks(masks, q.shape[0])
        off = np.zeros(q.shape[0] + 1, np.int32)
        nl, total = ctypes.c_int(), ctypes.c_int()
        cap = max(1024, 4 * q.shape[0])
        while True:
            out = (_LfDmatch * cap)()
            rc = self.lib.lf_matcher_radius_match(self.fe.h, _ptr(q), q.shape[0], ctypes.c_float(maxDistance), arr, int(bool(compactResult)), _ptr(off),
                                                  out, cap, ctypes.byref(nl), ctypes.byref(total))
            if rc == -2 and total.value > cap:          # LF_ERR_CAPACITY: size and repeat
                cap = total.value
                continue
            self.fe._check(rc)
            return [self._list(out, int(off[i]), int(off[i + 1])) for i in range(nl.value)]
