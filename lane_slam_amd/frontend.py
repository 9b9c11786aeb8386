"""Batch front end: the arithmetic of

    src/line_detector/src/line_detector_node.py:163-213      (processImage_)
    src/ground_projection/src/ground_projection_node.py:55-65 (lineseglist_cb)
    src/line_sanity/src/line_sanity_node.py:48-72             (processSegmentList)
    src/line_descriptor/src/binary_descriptor_custom.cpp:524-687 (BinaryDescriptor::compute)
    src/line_descriptor/src/binary_descriptor_matcher.cpp:197-254 (BinaryDescriptorMatcher::match)

(paths relative to /root/reference) for a whole batch of camera frames per call, on one
MI355X through liblanefront.so.  Inputs and outputs are numpy arrays (host) or raw device
pointers / torch CUDA tensors (device resident, no copies).
"""
import ctypes

import numpy as np

from . import _lib
from .config import LfConfig, default_config, fill_struct, work_size


class LanefrontError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "lanefront error %d: %s" % (code, msg))
        self.code = code


class Segments(object):
    """Struct-of-arrays SegmentList of a batch (numpy, host).  Field meaning: include/lanefront.h."""

    __slots__ = ("n", "frame_offset", "lines", "normals", "color", "pixels_normalized", "ground", "keep",
                 "desc", "code")

    def frame(self, f):
        a, b = int(self.frame_offset[f]), int(self.frame_offset[f + 1])
        out = Segments()
        out.n = b - a
        out.frame_offset = np.array([0, b - a], np.int32)
        for k in ("lines", "normals", "color", "pixels_normalized", "ground", "keep", "desc", "code"):
            v = getattr(self, k)
            setattr(out, k, None if v is None else v[a:b])
        return out


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class FrontEnd(object):
    def __init__(self, cfg=None, device=0, max_frames=1, max_lines_per_color=512):
        self.lib = _lib.load()
        self.cfg = default_config("parity") if cfg is None else cfg
        self.c = fill_struct(LfConfig(), self.cfg)
        self.max_frames = int(max_frames)
        self.cap_lines = int(max_lines_per_color)
        self.rows, self.cols = work_size(self.cfg)
        self.in_rows, self.in_cols = self.cfg["in_size"]
        self.h = ctypes.c_void_p()
        rc = self.lib.lf_create(ctypes.byref(self.c), int(device), self.max_frames, self.cap_lines, ctypes.byref(self.h))
        if rc != 0:
            msg = self.lib.lf_last_error(None).decode()
            self.h = None
            raise LanefrontError(rc, msg)
        r, c = ctypes.c_int(), ctypes.c_int()
        self.lib.lf_lsd_size(self.h, ctypes.byref(r), ctypes.byref(c))
        self.lsd_rows, self.lsd_cols = r.value, c.value

    def close(self):
        if getattr(self, "h", None):
            self.lib.lf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise LanefrontError(rc, self.lib.lf_last_error(self.h).decode())

    @property
    def capacity(self):
        return self.max_frames * 3 * self.cap_lines

    def synchronize(self):
        self._check(self.lib.lf_synchronize(self.h))

    def stream_ptr(self):
        """The handle's hipStream_t as an integer (for torch.cuda.ExternalStream and event-based ordering)."""
        p = ctypes.c_void_p()
        self._check(self.lib.lf_get_stream(self.h, ctypes.byref(p)))
        return p.value or 0

    # ------------------------------------------------------------------ host arrays
    def process_batch(self, frames, describe=True, n_frames=None):
        """frames: uint8 (n, in_rows, in_cols, 3) BGR on the host -- or, with n_frames given, the device
        address of such a block (e.g. frames_buffer() filled by decode_jpeg_batch).  Returns a host `Segments`."""
        on_device = n_frames is not None
        if on_device:
            n = int(n_frames)
            frames_arg = ctypes.c_void_p(int(frames))
        else:
            frames = np.ascontiguousarray(frames, dtype=np.uint8)
            if frames.ndim == 3:
                frames = frames[None]
            n = frames.shape[0]
            if frames.shape[1:] != (self.in_rows, self.in_cols, 3):
                raise ValueError("frames must be (n,%d,%d,3), got %r" % (self.in_rows, self.in_cols, frames.shape))
            frames_arg = _ptr(frames)
        cap = n * 3 * self.cap_lines
        out = Segments()
        out.frame_offset = np.zeros(n + 1, np.int32)
        out.lines = np.empty((cap, 4), np.float32)
        out.normals = np.empty((cap, 2), np.float32)
        out.color = np.empty(cap, np.uint8)
        out.pixels_normalized = np.empty((cap, 4), np.float32)
        out.ground = np.empty((cap, 4), np.float64)
        out.keep = np.empty(cap, np.uint8)
        out.desc = np.empty((cap, 72), np.float32) if describe else None
        out.code = np.empty((cap, 32), np.uint8) if describe else None
        s = _lib.LfSegments()
        s.capacity = cap
        for k in ("frame_offset", "lines", "normals", "color", "pixels_normalized", "ground", "keep"):
            setattr(s, k, getattr(out, k).ctypes.data)
        if describe:
            s.desc, s.code = out.desc.ctypes.data, out.code.ctypes.data
        total = ctypes.c_int()
        self._check(self.lib.lf_process_batch(self.h, frames_arg, n, int(on_device), ctypes.byref(s), 0, int(bool(describe)),
                                              ctypes.byref(total)))
        t = total.value
        out.n = t
        for k in ("lines", "normals", "color", "pixels_normalized", "ground", "keep", "desc", "code"):
            v = getattr(out, k)
            if v is not None:
                setattr(out, k, v[:t])
        return out

    # ------------------------------------------------------------------ device resident
    def process_batch_device(self, frames_ptr, n_frames, out_ptrs, capacity, describe=True):
        """frames_ptr: device address of uint8 (n, in_rows, in_cols, 3).  out_ptrs: dict of device
        addresses for any of frame_offset, lines, normals, color, pixels_normalized, ground, keep,
        desc, code (capacity segments each).  Returns the total segment count."""
        s = _lib.LfSegments()
        s.capacity = int(capacity)
        for k, v in out_ptrs.items():
            setattr(s, k, int(v))
        total = ctypes.c_int()
        self._check(self.lib.lf_process_batch(self.h, ctypes.c_void_p(int(frames_ptr)), int(n_frames), 1,
                                              ctypes.byref(s), 1, int(bool(describe)), ctypes.byref(total)))
        return total.value

    def submit_device(self, frames_ptr, n_frames, out_ptrs, capacity, describe=True):
        """Queue a batch (device pointers as in process_batch_device) and return immediately."""
        s = _lib.LfSegments()
        s.capacity = int(capacity)
        for k, v in out_ptrs.items():
            setattr(s, k, int(v))
        self._check(self.lib.lf_process_batch_async(self.h, ctypes.c_void_p(int(frames_ptr)), int(n_frames), 1,
                                                    ctypes.byref(s), int(bool(describe))))

    def submit_host(self, frames_host_ptr, n_frames, out_ptrs, capacity, describe=True):
        """Like submit_device, but the frames are in HOST memory (pinned, for the copy to be asynchronous): the H2D copy
        into the handle's staging buffer is queued on the handle's stream in front of the kernels.  The host block must
        stay valid until wait() returns."""
        s = _lib.LfSegments()
        s.capacity = int(capacity)
        for k, v in out_ptrs.items():
            setattr(s, k, int(v))
        self._check(self.lib.lf_process_batch_async(self.h, ctypes.c_void_p(int(frames_host_ptr)), int(n_frames), 0,
                                                    ctypes.byref(s), int(bool(describe))))

    def set_detector(self, detector, params=None):
        """Which detector process_batch / submit_* run: "lsd" (the reference's, default) or "edlines" (EDLines on the gray
        working image + the colour masks: include/lanefront.h, lf_set_detector).  params: edlines_params(...) or None."""
        if detector not in _lib.DETECTORS:
            raise ValueError("detector must be one of %r" % (sorted(_lib.DETECTORS),))
        self._check(self.lib.lf_set_detector(self.h, _lib.DETECTORS[detector], ctypes.byref(params) if params is not None else None))

    def detector_failures(self):
        """Frames of the last completed batch on which the EDLines detector gave up (they have no segments)."""
        return int(self.lib.lf_detector_failures(self.h))

    def suggested_depth(self):
        """Batches (handles) worth keeping in flight for the content this handle saw last (lf_suggested_depth): 8 on lane
        frames, 18 on busy camera content.  A throughput hint only."""
        return int(self.lib.lf_suggested_depth(self.h))

    def lsd_list_capacity(self):
        """(entries per problem the LSD stages' lists hold, times they were grown): lf_lsd_list_capacity.  A batch with a problem
        that needs more is run a second time by wait() after the lists were reallocated; results never depend on it."""
        e, g = ctypes.c_int(0), ctypes.c_int(0)
        self._check(self.lib.lf_lsd_list_capacity(self.h, ctypes.byref(e), ctypes.byref(g)))
        return int(e.value), int(g.value)

    def wait(self):
        """Block until the queued batch is complete; returns its segment count."""
        total = ctypes.c_int()
        self._check(self.lib.lf_wait(self.h, ctypes.byref(total)))
        return total.value

    # ------------------------------------------------------------------ association
    def set_tie_rule(self, rule):
        """Which of several equally near map codes `associate` returns: "mihasher" (the default: the one the reference's
        Mihasher::query discovers first, binary_descriptor_matcher.cpp:635-753) or "lowest" (lowest index, one pass)."""
        if rule not in _lib.TIE_RULES:
            raise ValueError("tie rule must be one of %r" % (sorted(_lib.TIE_RULES),))
        self._check(self.lib.lf_set_tie_rule(self.h, _lib.TIE_RULES[rule]))

    def associate(self, query_codes, map_codes):
        """Exact Hamming NN (binary_descriptor_matcher.cpp:197-254).  Returns (idx int32, dist float32);
        idx == -1 where the nearest map entry is farther than 128 bits or the map is empty.  Equally near codes: see
        set_tie_rule."""
        q = np.ascontiguousarray(query_codes, dtype=np.uint8).reshape(-1, 32)
        m = np.ascontiguousarray(map_codes, dtype=np.uint8).reshape(-1, 32)
        idx = np.empty(q.shape[0], np.int32)
        dist = np.empty(q.shape[0], np.float32)
        self._check(self.lib.lf_associate(self.h, _ptr(q), q.shape[0], _ptr(m), m.shape[0], _ptr(idx), _ptr(dist), 0))
        return idx, dist

    def associate_device(self, q_ptr, nq, m_ptr, nm, idx_ptr, dist_ptr):
        self._check(self.lib.lf_associate(self.h, ctypes.c_void_p(int(q_ptr)), int(nq), ctypes.c_void_p(int(m_ptr)),
                                          int(nm), ctypes.c_void_p(int(idx_ptr)), ctypes.c_void_p(int(dist_ptr)), 1))

    def select_queries(self, query_codes, mask):
        """The matcher's per-query mask (ref: binary_descriptor_matcher.cpp:231-235): (codes of the queries whose mask byte is not
        0, their row numbers = DMatch.queryIdx).  Feed the codes to associate / knn_match / radius_match."""
        q = np.ascontiguousarray(query_codes, dtype=np.uint8).reshape(-1, 32)
        m = np.ascontiguousarray(mask, dtype=np.uint8).reshape(-1)
        if m.shape[0] != q.shape[0]:
            raise ValueError("mask must have one byte per query")
        sel = np.empty_like(q)
        qi = np.empty(q.shape[0], np.int32)
        n = ctypes.c_int()
        self._check(self.lib.lf_select_queries(self.h, _ptr(q), q.shape[0], _ptr(m), _ptr(sel), _ptr(qi), ctypes.byref(n), 0))
        return sel[:n.value].copy(), qi[:n.value].copy()

    def knn_match(self, query_codes, map_codes, k):
        """BinaryDescriptorMatcher::knnMatch (ref: binary_descriptor_matcher.cpp:258-335): (idx [nq, k] int32, dist [nq, k]
        float32), nearest first, within 128 bits; -1 where there are fewer than k."""
        q = np.ascontiguousarray(query_codes, dtype=np.uint8).reshape(-1, 32)
        m = np.ascontiguousarray(map_codes, dtype=np.uint8).reshape(-1, 32)
        idx = np.empty((q.shape[0], int(k)), np.int32)
        dist = np.empty((q.shape[0], int(k)), np.float32)
        self._check(self.lib.lf_knn_match(self.h, _ptr(q), q.shape[0], _ptr(m), m.shape[0], int(k), _ptr(idx), _ptr(dist), 0))
        return idx, dist

    def radius_match(self, query_codes, map_codes, max_distance):
        """BinaryDescriptorMatcher::radiusMatch (ref: binary_descriptor_matcher.cpp:428-504): (offsets [nq + 1], idx, dist) --
        the matches of query i are idx[offsets[i]:offsets[i + 1]], nearest first."""
        q = np.ascontiguousarray(query_codes, dtype=np.uint8).reshape(-1, 32)
        m = np.ascontiguousarray(map_codes, dtype=np.uint8).reshape(-1, 32)
        offsets = np.zeros(q.shape[0] + 1, np.int32)
        total = ctypes.c_int()
        cap = max(1024, 4 * q.shape[0])
        while True:
            idx, dist = np.empty(cap, np.int32), np.empty(cap, np.float32)
            rc = self.lib.lf_radius_match(self.h, _ptr(q), q.shape[0], _ptr(m), m.shape[0], float(max_distance), _ptr(offsets), _ptr(idx),
                                          _ptr(dist), cap, ctypes.byref(total), 0)
            if rc == -2 and total.value > cap:          # LF_ERR_CAPACITY: offsets and total are complete, size and repeat
                cap = total.value
                continue
            self._check(rc)
            return offsets, idx[:total.value], dist[:total.value]

    def associate_float(self, query_desc, map_desc):
        q = np.ascontiguousarray(query_desc, dtype=np.float32).reshape(-1, 72)
        m = np.ascontiguousarray(map_desc, dtype=np.float32).reshape(-1, 72)
        idx = np.empty(q.shape[0], np.int32)
        dist = np.empty(q.shape[0], np.float32)
        self._check(self.lib.lf_associate_float(self.h, _ptr(q), q.shape[0], _ptr(m), m.shape[0], _ptr(idx), _ptr(dist), 0))
        return idx, dist

    def kmeans(self, bgr_points, init_centers, max_iter=25, tol=1e-4):
        """Lloyd's k-means from an explicit init on [N, 3] u8 B, G, R points (anti_instagram/kmeans.py:24-26, i.e.
        sklearn.cluster.KMeans(n_clusters, max_iter=25, init=<array>)).  Returns (centers [k, 3] f64, counts [k] i64,
        inertia, n_iter)."""
        pts = np.ascontiguousarray(bgr_points, np.uint8).reshape(-1, 3)
        init = np.ascontiguousarray(init_centers, np.float64).reshape(-1, 3)
        k = init.shape[0]
        centers = np.zeros((k, 3), np.float64)
        counts = np.zeros(k, np.int64)
        inertia = ctypes.c_double()
        n_iter = ctypes.c_int()
        self._check(self.lib.lf_kmeans(self.h, _ptr(pts), pts.shape[0], 0, k, _ptr(init), int(max_iter), float(tol), _ptr(centers),
                                        _ptr(counts), ctypes.byref(inertia), ctypes.byref(n_iter)))
        return centers, counts, inertia.value, n_iter.value

    # ------------------------------------------------------------------ EDLines / multi-octave KeyLines
    def set_descriptor_params(self, num_of_octave=None, width_of_band=None, reduction_ratio=None, ksize=None):
        """BinaryDescriptor's setters (setNumOfOctaves / setWidthOfBand / setReductionRatio, Params::ksize_;
        ref: src/line_descriptor/src/binary_descriptor_custom.cpp:119-187) on this handle: lf_set_descriptor_params.  Arguments left at
        None keep their value.  Returns the parameters now in force as a dict."""
        p = _lib.LfDescriptorParams()
        self._check(self.lib.lf_get_descriptor_params(self.h, ctypes.byref(p)))
        for k, v in (("num_of_octave", num_of_octave), ("width_of_band", width_of_band), ("reduction_ratio", reduction_ratio), ("ksize", ksize)):
            if v is not None:
                setattr(p, k, int(v))
        self._check(self.lib.lf_set_descriptor_params(self.h, ctypes.byref(p)))
        return self.descriptor_params()

    def descriptor_params(self):
        p = _lib.LfDescriptorParams()
        self._check(self.lib.lf_get_descriptor_params(self.h, ctypes.byref(p)))
        return {"num_of_octave": p.num_of_octave, "width_of_band": p.width_of_band, "reduction_ratio": p.reduction_ratio, "ksize": p.ksize}

    def edlines_params(self, **kw):
        """EDLineDetector's defaults (ref: binary_descriptor_custom.cpp:1374-1385), optionally overridden."""
        p = _lib.LfEdlinesParams()
        self.lib.lf_edlines_default_params(ctypes.byref(p))
        for k, v in kw.items():
            setattr(p, k, v)
        return p

    def keylines_batch(self, images, n_octaves=1, describe=True, params=None, gray=False, capacity=None, masks=None):
        """BinaryDescriptor::operator() with drawn KeyLines (ref: binary_descriptor_custom.cpp:263-301): EDLines over
        n_octaves octaves + LBD on the detector's gradients.  images: raw camera frames (n, in_rows, in_cols, 3) BGR, or
        with gray=True working-size gray images (n, rows, cols) u8.  Returns a dict of per-KeyLine numpy arrays (KeyLine
        fields, include/lanefront.h) + 'n', 'frame_offset' and 'frame_status'.
        masks: (n, rows, cols) u8 of the working size = detect's mask argument (:509-519): a KeyLine with both end points on zero
        pixels is erased, by the reference's loop as written (no step back after an erase: see include/lanefront.h)."""
        images = np.ascontiguousarray(images, np.uint8)
        want = (self.rows, self.cols) if gray else (self.in_rows, self.in_cols, 3)
        if images.ndim == len(want):
            images = images[None]
        if images.shape[1:] != want:
            raise ValueError("images must be (n,%s), got %r" % (",".join(map(str, want)), images.shape))
        n = images.shape[0]
        cap = int(capacity) if capacity else n * 2048
        out = {"frame_offset": np.zeros(n + 1, np.int32)}
        s = _lib.LfKeylines()
        s.capacity = cap
        s.frame_offset = out["frame_offset"].ctypes.data
        for k, dt, c in _lib.KEYLINE_FIELDS:
            if k in ("desc", "code") and not describe:
                continue
            out[k] = np.zeros((cap, c) if c > 1 else cap, np.dtype(dt))
            setattr(s, k, out[k].ctypes.data)
        total = ctypes.c_int()
        status = np.zeros(n, np.int32)
        if masks is not None:
            masks = np.ascontiguousarray(masks, np.uint8)
            if masks.ndim == 2:
                masks = masks[None]
            if masks.shape != (n, self.rows, self.cols):
                raise ValueError("masks must be (%d,%d,%d), got %r" % (n, self.rows, self.cols, masks.shape))
            self._check(self.lib.lf_keylines_batch_masked(self.h, _ptr(images), n, 1 if gray else 0, 0, int(n_octaves),
                                                          ctypes.byref(params) if params is not None else None, _ptr(masks), 0, ctypes.byref(s), 0,
                                                          int(bool(describe)), ctypes.byref(total), _ptr(status)))
        else:
            self._check(self.lib.lf_keylines_batch(self.h, _ptr(images), n, 1 if gray else 0, 0, int(n_octaves),
                                                   ctypes.byref(params) if params is not None else None, ctypes.byref(s), 0,
                                                   int(bool(describe)), ctypes.byref(total), _ptr(status)))
        t = total.value
        for k, _, _ in _lib.KEYLINE_FIELDS:
            if k in out:
                out[k] = out[k][:t]
        out["n"] = t
        out["frame_status"] = status
        return out

    def lsd_options(self, **kw):
        """LSDDetectorC::LSDOptions (ref: descriptor_custom.hpp:906-916) with cv::createLineSegmentDetector()'s defaults and min_length 0;
        keywords: refine, scale, sigma_scale, quant, ang_th, log_eps, density_th, n_bins, min_length."""
        o = _lib.LfLsdOptions()
        self.lib.lf_lsd_default_options(ctypes.byref(o))
        for k, v in kw.items():
            if not hasattr(o, k):
                raise ValueError("unknown LSD option %r" % (k,))
            setattr(o, k, v)
        return o

    def lsd_keylines_batch(self, images, n_octaves=1, describe=True, gray=False, capacity=None, options=None, masks=None):
        """LSDDetectorC::detect over n_octaves pyramid levels (ref: LSDDetector_custom.cpp:130-215) + BinaryDescriptor::compute:
        the same dict as keylines_batch (no 'frame_status'; 'salience' is 0).  images as in keylines_batch.
        options (lsd_options(...)): the fork's detect(..., LSDOptions, mask) / detectFast overloads (:218-438) -- the detector's
        parameters and the min_length filter; masks: (n, rows, cols) u8 of the working size -- KeyLines with both end points on zero
        mask pixels are erased (:203-213)."""
        images = np.ascontiguousarray(images, np.uint8)
        want = (self.rows, self.cols) if gray else (self.in_rows, self.in_cols, 3)
        if images.ndim == len(want):
            images = images[None]
        if images.shape[1:] != want:
            raise ValueError("images must be (n,%s), got %r" % (",".join(map(str, want)), images.shape))
        n = images.shape[0]
        cap = int(capacity) if capacity else n * 2048
        out = {"frame_offset": np.zeros(n + 1, np.int32)}
        s = _lib.LfKeylines()
        s.capacity = cap
        s.frame_offset = out["frame_offset"].ctypes.data
        for k, dt, c in _lib.KEYLINE_FIELDS:
            if k in ("desc", "code") and not describe:
                continue
            out[k] = np.zeros((cap, c) if c > 1 else cap, np.dtype(dt))
            setattr(s, k, out[k].ctypes.data)
        total = ctypes.c_int()
        if masks is not None:
            masks = np.ascontiguousarray(masks, np.uint8)
            if masks.ndim == 2:
                masks = masks[None]
            if masks.shape != (n, self.rows, self.cols):
                raise ValueError("masks must be (%d,%d,%d), got %r" % (n, self.rows, self.cols, masks.shape))
        self._check(self.lib.lf_lsd_keylines_batch_ex(self.h, _ptr(images), n, 1 if gray else 0, 0, int(n_octaves),
                                                      ctypes.byref(options) if options is not None else None,
                                                      _ptr(masks) if masks is not None else None, 0, ctypes.byref(s), 0,
                                                      int(bool(describe)), ctypes.byref(total)))
        t = total.value
        for k, _, _ in _lib.KEYLINE_FIELDS:
            if k in out:
                out[k] = out[k][:t]
        out["n"] = t
        return out

    def keylines_submit_device(self, images_ptr, n_frames, out_ptrs, capacity, n_octaves=1, describe=True, params=None, gray=False):
        """Queue lf_keylines_batch_async: images_ptr = device address of the frames ((n, in_rows, in_cols, 3) BGR, or with
        gray=True (n, rows, cols) u8), out_ptrs = {KeyLine field or "frame_offset": device address}.  wait() returns the
        KeyLine total; keylines_frame_status() the per-frame status afterwards."""
        s = _lib.LfKeylines()
        s.capacity = int(capacity)
        for k, v in out_ptrs.items():
            setattr(s, k, int(v))
        self._check(self.lib.lf_keylines_batch_async(self.h, ctypes.c_void_p(int(images_ptr)), int(n_frames), 1 if gray else 0, int(n_octaves),
                                                     ctypes.byref(params) if params is not None else None, ctypes.byref(s), int(bool(describe))))

    def keylines_frame_status(self, n_frames):
        st = np.zeros(int(n_frames), np.int32)
        self._check(self.lib.lf_keylines_frame_status(self.h, _ptr(st), int(n_frames)))
        return st

    def describe_keylines(self, gray, line_frame, in_octave, angle, num_pixels, octave):
        """BinaryDescriptor::compute on GIVEN KeyLines (ref: binary_descriptor_custom.cpp:524-687; pyramid of
        computeGaussianPyramid :350-371).  gray: (n_frames, rows, cols) u8.  Returns (desc [n, 72], code [n, 32])."""
        gray = np.ascontiguousarray(gray, np.uint8)
        if gray.ndim == 2:
            gray = gray[None]
        if gray.shape[1:] != (self.rows, self.cols):
            raise ValueError("gray must be (n,%d,%d)" % (self.rows, self.cols))
        fr = np.ascontiguousarray(line_frame, np.int32)
        io = np.ascontiguousarray(in_octave, np.float32).reshape(-1, 4)
        ang = np.ascontiguousarray(angle, np.float32)
        npx = np.ascontiguousarray(num_pixels, np.int32)
        octv = np.ascontiguousarray(octave, np.int32)
        n = io.shape[0]
        desc, code = np.zeros((n, 72), np.float32), np.zeros((n, 32), np.uint8)
        self._check(self.lib.lf_describe_keylines(self.h, _ptr(gray), gray.shape[0], _ptr(fr), _ptr(io), _ptr(ang), _ptr(npx), _ptr(octv), n,
                                                  _ptr(desc), _ptr(code), 0))
        return desc, code

    def keylines_fetch(self, octave, what, n_frames):
        """Intermediate buffer `what` (include/lanefront.h, lf_keylines_debug_fetch) of the last keylines_batch."""
        dims = np.zeros(5, np.int32)
        self._check(self.lib.lf_keylines_debug_fetch(self.h, int(octave), int(what), None, 0, _ptr(dims)))
        H, W, cap, max_edges, max_lines = (int(v) for v in dims)
        shape, dt = {0: ((H, W), np.uint8), 1: ((H, W), np.uint32), 2: ((H, W), np.uint16), 3: ((cap,), np.uint32), 4: ((2 * cap,), np.uint32),
                     5: ((max_edges + 2,), np.uint32), 6: ((4,), np.int32), 7: ((max_lines, 4), np.float32), 8: ((max_lines,), np.float64),
                     9: ((max_lines,), np.float32), 10: ((max_lines,), np.int32), 11: ((max_lines,), np.float32), 12: ((H, W), np.uint8)}[what]
        a = np.empty((n_frames,) + shape, dt)
        self._check(self.lib.lf_keylines_debug_fetch(self.h, int(octave), int(what), _ptr(a), a.nbytes, None))
        return a

    # ------------------------------------------------------------------ host ingest (JPEG)
    def decode_jpeg_batch(self, streams, rows=None, cols=None, n_threads=0, device_ptr=None, entropy="gpu"):
        """Decode a list of JPEG byte strings (what CompressedImage.data carries) into BGR frames --
        the batched form of duckietown_utils.jpg.image_cv_from_jpg = cv2.imdecode(data, IMREAD_COLOR)
        (ref: duckietown_utils/jpg.py:21-31).  entropy="gpu" (default): the host parses headers only, Huffman decoding
        and everything after it run on the GPU (lf_jpeg_decode_batch_gpu); entropy="host": Huffman decoding on n_threads
        host threads (lf_jpeg_decode_batch).  Same output either way.

        device_ptr None: returns (frames u8 [n, rows, cols, 3] on the host, status int32 [n]);
        device_ptr = a device address: frames are written there (asynchronously on the handle's stream)
        and only status is returned.  Frames whose status is not 0 are zeros (the reference drops them)."""
        n = len(streams)
        rows = self.cfg["in_size"][0] if rows is None else rows
        cols = self.cfg["in_size"][1] if cols is None else cols
        # bytes objects are handed over as they are (no copy): at tens of thousands of frames per second even one extra
        # pass over the streams on the Python side would show
        keep = [b if isinstance(b, bytes) else bytes(b) for b in streams]
        ptrs = ctypes.cast((ctypes.c_char_p * n)(*[b if len(b) else None for b in keep]), ctypes.POINTER(ctypes.c_void_p))
        sizes = (ctypes.c_size_t * n)(*[len(b) for b in keep])
        status = np.zeros(n, np.int32)
        st = status.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
        if entropy not in ("gpu", "host"):
            raise ValueError("entropy must be 'gpu' or 'host'")
        fn = self.lib.lf_jpeg_decode_batch_gpu if entropy == "gpu" else self.lib.lf_jpeg_decode_batch
        if device_ptr is None:
            out = np.empty((n, rows, cols, 3), np.uint8)
            self._check(fn(self.h, ptrs, sizes, n, rows, cols, _ptr(out), 0, n_threads, st))
            return out, status
        self._check(fn(self.h, ptrs, sizes, n, rows, cols, ctypes.c_void_p(int(device_ptr)), 1, n_threads, st))
        return status

    def decode_jpeg_batch_async(self, streams, device_ptr=None, rows=None, cols=None, n_threads=0, for_detect=False):
        """The queued form of decode_jpeg_batch(entropy="gpu") (lf_jpeg_decode_batch_gpu_async): headers are parsed and the entropy-coded
        bytes staged before the call returns, everything else runs behind it on the handle's stream -- follow it with submit_device on
        the same buffer.  device_ptr None: the handle's own frame buffer.  The per-frame status comes from jpeg_status().  Returns the
        device address the frames are written to.

        for_detect=True (lf_jpeg_decode_for_detect_async): into the handle's own buffer, and of every frame only the rows the front end
        reads -- from the crop line on; what lies above is never looked at by the detector and is not decoded past the Huffman stage."""
        n = len(streams)
        rows = self.cfg["in_size"][0] if rows is None else rows
        cols = self.cfg["in_size"][1] if cols is None else cols
        own = self.frames_buffer()[0]
        if device_ptr is None:
            device_ptr = own
        keep = [b if isinstance(b, bytes) else bytes(b) for b in streams]
        ptrs = ctypes.cast((ctypes.c_char_p * n)(*[b if len(b) else None for b in keep]), ctypes.POINTER(ctypes.c_void_p))
        sizes = (ctypes.c_size_t * n)(*[len(b) for b in keep])
        if for_detect:
            if int(device_ptr) != own or (rows, cols) != tuple(self.cfg["in_size"]):
                raise ValueError("for_detect decodes frames of the configured size into the handle's own buffer")
            self._check(self.lib.lf_jpeg_decode_for_detect_async(self.h, ptrs, sizes, n, n_threads))
        else:
            self._check(self.lib.lf_jpeg_decode_batch_gpu_async(self.h, ptrs, sizes, n, rows, cols, ctypes.c_void_p(int(device_ptr)), n_threads))
        self._jpeg_queued = n
        return int(device_ptr)

    def jpeg_status(self):
        """Per-frame status (int32 [n]) of the batch decode_jpeg_batch_async queued last; waits for that decode alone."""
        n = getattr(self, "_jpeg_queued", 0)
        status = np.zeros(n, np.int32)
        self._check(self.lib.lf_jpeg_status(self.h, status.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), n, None))
        return status

    def frames_buffer(self):
        """(device address, bytes) of the handle's own input staging buffer ([max_frames][in_rows][in_cols][3])."""
        p, nb = ctypes.c_void_p(), ctypes.c_size_t()
        self._check(self.lib.lf_frames_buffer(self.h, ctypes.byref(p), ctypes.byref(nb)))
        return p.value, nb.value

    # ------------------------------------------------------------------ introspection
    def fetch(self, buffer_id, n_frames):
        """Intermediate buffer of the last batch (tests / debugging)."""
        P, Ps = self.rows * self.cols, self.lsd_rows * self.lsd_cols
        shapes = {
            _lib.LF_BUF_BGR: ((n_frames, self.rows, self.cols, 3), np.uint8),
            _lib.LF_BUF_MASKS: ((n_frames, 3, self.rows, self.cols), np.uint8),
            _lib.LF_BUF_EDGES: ((n_frames, self.rows, self.cols), np.uint8),
            _lib.LF_BUF_LSD_ANGLE: ((n_frames, 3, self.lsd_rows, self.lsd_cols), np.float32),
            _lib.LF_BUF_LSD_MODGRAD: ((n_frames, 3, self.lsd_rows, self.lsd_cols), np.float64),
            _lib.LF_BUF_LSD_ORDER: ((n_frames, 3, Ps), np.uint32),
            _lib.LF_BUF_LSD_NORDER: ((n_frames, 3), np.int32),
            _lib.LF_BUF_LBD_DX: ((n_frames, self.rows, self.cols), np.int16),
            _lib.LF_BUF_LBD_DY: ((n_frames, self.rows, self.cols), np.int16),
            _lib.LF_BUF_LSD_COUNTS: ((n_frames, 3), np.int32),
            _lib.LF_BUF_LSD_NLOW: ((n_frames, 3), np.int32),
            _lib.LF_BUF_LSD_SCRATCH: ((n_frames, 3, int(self.lib.lf_lsd_scratch_stride(self.h))), np.uint32),     # follows the list capacity (k_lsd_grow.hip lsd_grow_reg_stride)
        }
        shape, dt = shapes[buffer_id]
        a = np.empty(shape, dt)
        self._check(self.lib.lf_debug_fetch(self.h, buffer_id, _ptr(a), a.nbytes))
        return a

    def lsd_binary(self, img, cap=8192):
        """LSD stages alone on a binary working-size image (tests / diagnosis)."""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        lines = np.empty((cap, 4), np.float32)
        n = ctypes.c_int()
        self._check(self.lib.lf_debug_lsd_binary(self.h, _ptr(img), img.shape[0], img.shape[1], _ptr(lines), cap, ctypes.byref(n)))
        return lines[:n.value].copy()

    def set_profiling(self, on):
        self._check(self.lib.lf_set_profiling(self.h, int(bool(on))))

    def reset_timing(self):
        self._check(self.lib.lf_reset_timing(self.h))

    def timing(self):
        """{stage name: (ms accumulated, launches)} measured with HIP events on the handle's stream."""
        ms = np.zeros(_lib.LF_N_STAGES, np.float64)
        ln = np.zeros(_lib.LF_N_STAGES, np.int32)
        self._check(self.lib.lf_get_timing(self.h, _ptr(ms), _ptr(ln), _lib.LF_N_STAGES))
        return {self.lib.lf_stage_name(i).decode(): (float(ms[i]), int(ln[i])) for i in range(_lib.LF_N_STAGES)}
