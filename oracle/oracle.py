"""ORACLE -- TEST INFRASTRUCTURE ONLY.

ctypes wrapper around oracle/_build/liblforacle.so (the plain-C CPU restatement of
the reference's line-feature front end).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; nothing under lane_slam_amd/
does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liblforacle.so")


def build(force=False):
    """Compile the oracle with gcc (idempotent)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h", ".cpp"))]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs)):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class LfoConfig(ctypes.Structure):
    """ctypes mirror of `lfo_config` (oracle/lf_oracle.h)."""
    _fields_ = [
        ("in_rows", ctypes.c_int32), ("in_cols", ctypes.c_int32),
        ("img_rows", ctypes.c_int32), ("img_cols", ctypes.c_int32),
        ("top_cutoff", ctypes.c_int32),
        ("ai_scale", ctypes.c_float * 3), ("ai_shift", ctypes.c_float * 3),
        ("hsv_lo", (ctypes.c_int32 * 3) * 4), ("hsv_hi", (ctypes.c_int32 * 3) * 4),
        ("dilation_kernel_size", ctypes.c_int32),
        ("canny_lo", ctypes.c_double), ("canny_hi", ctypes.c_double),
        ("lsd_refine", ctypes.c_int32), ("lsd_n_bins", ctypes.c_int32),
        ("lsd_scale", ctypes.c_double), ("lsd_sigma_scale", ctypes.c_double),
        ("lsd_quant", ctypes.c_double), ("lsd_ang_th", ctypes.c_double),
        ("lsd_log_eps", ctypes.c_double), ("lsd_density_th", ctypes.c_double),
        ("H", ctypes.c_double * 9), ("K", ctypes.c_double * 9), ("D", ctypes.c_double * 5),
        ("R", ctypes.c_double * 9), ("P", ctypes.c_double * 12),
        ("cam_w", ctypes.c_int32), ("cam_h", ctypes.c_int32),
        ("lanewidth", ctypes.c_double), ("linewidth_white", ctypes.c_double),
        ("linewidth_yellow", ctypes.c_double), ("d_min", ctypes.c_double),
        ("d_max", ctypes.c_double), ("phi_min", ctypes.c_double), ("phi_max", ctypes.c_double),
        ("lsd_seed_order", ctypes.c_int32), ("reserved0", ctypes.c_int32),
    ]


class _FrameOut(ctypes.Structure):
    _fields_ = [
        ("n", ctypes.c_int32), ("n_color", ctypes.c_int32 * 3),
        ("lines", ctypes.c_void_p), ("normals", ctypes.c_void_p), ("color", ctypes.c_void_p),
        ("pixels_normalized", ctypes.c_void_p), ("ground", ctypes.c_void_p),
        ("keep", ctypes.c_void_p), ("desc", ctypes.c_void_p), ("code", ctypes.c_void_p),
    ]


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _struct_from_dict(cfg):
    s = LfoConfig()
    det = cfg["detector"]
    s.in_rows, s.in_cols = cfg["in_size"]
    s.img_rows, s.img_cols = cfg["img_size"]
    s.top_cutoff = cfg["top_cutoff"]
    for i in range(3):
        s.ai_scale[i] = float(cfg["ai_scale"][i])
        s.ai_shift[i] = float(cfg["ai_shift"][i])
    boxes = [("hsv_white1", "hsv_white2"), ("hsv_yellow1", "hsv_yellow2"),
             ("hsv_red1", "hsv_red2"), ("hsv_red3", "hsv_red4")]
    for k, (lo, hi) in enumerate(boxes):
        for ch in range(3):
            s.hsv_lo[k][ch] = int(det[lo][ch])
            s.hsv_hi[k][ch] = int(det[hi][ch])
    s.dilation_kernel_size = int(det["dilation_kernel_size"])
    s.canny_lo, s.canny_hi = float(det["canny_thresholds"][0]), float(det["canny_thresholds"][1])
    lsd = cfg["lsd"]
    s.lsd_refine, s.lsd_n_bins = int(lsd["refine"]), int(lsd["n_bins"])
    s.lsd_scale, s.lsd_sigma_scale = float(lsd["scale"]), float(lsd["sigma_scale"])
    s.lsd_quant, s.lsd_ang_th = float(lsd["quant"]), float(lsd["ang_th"])
    s.lsd_log_eps, s.lsd_density_th = float(lsd["log_eps"]), float(lsd["density_th"])
    for name, n in (("H", 9), ("K", 9), ("D", 5), ("R", 9), ("P", 12)):
        arr = getattr(s, name)
        for i in range(n):
            arr[i] = float(cfg[name][i])
    s.cam_h, s.cam_w = cfg["cam_size"]
    for k, v in cfg["sanity"].items():
        setattr(s, k, float(v))
    s.lsd_seed_order = {"opencv30": 0, "opencv32": 1}[lsd.get("seed_order", "opencv32")]
    return s


class Oracle(object):
    """CPU oracle bound to one configuration dict (lane_slam_amd.config.default_config layout)."""

    def __init__(self, cfg):
        build()
        self.lib = ctypes.CDLL(_SO)
        self.lib.lfo_lsd_detect.restype = ctypes.c_int
        self.lib.lfo_lsd_ll_angle.restype = ctypes.c_int
        self.lib.lfo_process_frame.restype = ctypes.c_int
        self.cfg = cfg
        self.c = _struct_from_dict(cfg)
        self.rows = cfg["img_size"][0] - cfg["top_cutoff"]
        self.cols = cfg["img_size"][1]

    # ---- image stages
    def preprocess(self, bgr_in):
        bgr_in = np.ascontiguousarray(bgr_in, dtype=np.uint8)
        assert bgr_in.shape == (self.c.in_rows, self.c.in_cols, 3)
        out = np.empty((self.rows, self.cols, 3), np.uint8)
        self.lib.lfo_preprocess(ctypes.byref(self.c), _p(bgr_in), _p(out))
        return out

    def bgr2hsv(self, bgr):
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        out = np.empty_like(bgr)
        self.lib.lfo_bgr2hsv(_p(bgr), bgr.size // 3, _p(out))
        return out

    def color_masks(self, hsv):
        hsv = np.ascontiguousarray(hsv, dtype=np.uint8)
        npix = hsv.size // 3
        out = np.empty((3,) + hsv.shape[:-1], np.uint8)
        self.lib.lfo_color_masks(ctypes.byref(self.c), _p(hsv), npix, _p(out))
        return out

    def dilate(self, bw, ksize=None):
        bw = np.ascontiguousarray(bw, dtype=np.uint8)
        out = np.empty_like(bw)
        k = self.c.dilation_kernel_size if ksize is None else ksize
        self.lib.lfo_dilate_ellipse(_p(bw), bw.shape[0], bw.shape[1], k, _p(out))
        return out

    def canny(self, bgr, lo=None, hi=None):
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        out = np.empty(bgr.shape[:2], np.uint8)
        lo = self.c.canny_lo if lo is None else lo
        hi = self.c.canny_hi if hi is None else hi
        self.lib.lfo_canny_bgr(_p(bgr), bgr.shape[0], bgr.shape[1],
                               ctypes.c_double(lo), ctypes.c_double(hi), _p(out))
        return out

    def bgr2gray(self, bgr):
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        out = np.empty(bgr.shape[:-1], np.uint8)
        self.lib.lfo_bgr2gray(_p(bgr), out.size, _p(out))
        return out

    def gaussian5(self, gray):
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        out = np.empty_like(gray)
        self.lib.lfo_gaussian5_u8(_p(gray), gray.shape[0], gray.shape[1], _p(out))
        return out

    def sobel3(self, gray):
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        dx = np.empty(gray.shape, np.int16)
        dy = np.empty(gray.shape, np.int16)
        self.lib.lfo_sobel3_s16(_p(gray), gray.shape[0], gray.shape[1], _p(dx), _p(dy))
        return dx, dy

    # ---- LSD
    def lsd_scaled_size(self, rows, cols):
        a, b = ctypes.c_int(), ctypes.c_int()
        self.lib.lfo_lsd_scaled_size(ctypes.byref(self.c), rows, cols, ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    def lsd_scaled_image(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        sr, sc = self.lsd_scaled_size(*img.shape)
        out = np.empty((sr, sc), np.float64)
        self.lib.lfo_lsd_scaled_image(ctypes.byref(self.c), _p(img), img.shape[0], img.shape[1], _p(out))
        return out

    def lsd_ll_angle(self, scaled):
        scaled = np.ascontiguousarray(scaled, dtype=np.float64)
        ang = np.empty_like(scaled)
        mod = np.empty_like(scaled)
        order = np.empty(scaled.size, np.int32)
        n = self.lib.lfo_lsd_ll_angle(ctypes.byref(self.c), _p(scaled), scaled.shape[0], scaled.shape[1],
                                      _p(ang), _p(mod), _p(order))
        return ang, mod, order[:n]

    def lsd(self, img, cap=4096, extra=False):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        lines = np.empty((cap, 4), np.float32)
        ex = np.empty((cap, 3), np.float64)
        n = self.lib.lfo_lsd_detect(ctypes.byref(self.c), _p(img), img.shape[0], img.shape[1],
                                    _p(lines), _p(ex), cap)
        return (lines[:n].copy(), ex[:n].copy()) if extra else lines[:n].copy()

    # ---- per segment
    def find_normals(self, bw, lines):
        bw = np.ascontiguousarray(bw, dtype=np.uint8)
        lines = np.array(lines, dtype=np.float32, order="C").reshape(-1, 4)
        n = lines.shape[0]
        normals = np.empty((n, 2), np.float64)
        centers = np.empty((n, 2), np.float32)
        self.lib.lfo_find_normals(_p(bw), bw.shape[0], bw.shape[1], _p(lines), n, _p(normals), _p(centers))
        return lines, normals, centers

    def normalize_lines(self, lines):
        lines = np.ascontiguousarray(lines, dtype=np.float32).reshape(-1, 4)
        out = np.empty_like(lines)
        self.lib.lfo_normalize_lines(ctypes.byref(self.c), _p(lines), lines.shape[0], _p(out))
        return out

    def ground_project(self, pn):
        pn = np.ascontiguousarray(pn, dtype=np.float32).reshape(-1, 4)
        out = np.empty((pn.shape[0], 4), np.float64)
        self.lib.lfo_ground_project(ctypes.byref(self.c), _p(pn), pn.shape[0], _p(out))
        return out

    def line_sanity(self, pts, color):
        pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 4)
        color = np.ascontiguousarray(color, dtype=np.uint8)
        n = pts.shape[0]
        keep = np.empty(n, np.uint8)
        dphil = np.empty((n, 3), np.float64)
        state = np.empty(n, np.int32)
        self.lib.lfo_line_sanity(ctypes.byref(self.c), _p(pts), _p(color), n, _p(keep), _p(dphil), _p(state))
        return keep, dphil, state

    # ---- LBD / matcher
    def keylines(self, lines, rows, cols):
        lines = np.ascontiguousarray(lines, dtype=np.float32).reshape(-1, 4)
        n = lines.shape[0]
        ext = np.empty((n, 4), np.float32)
        ang = np.empty(n, np.float32)
        npx = np.empty(n, np.int32)
        self.lib.lfo_keylines(_p(lines), n, rows, cols, _p(ext), _p(ang), _p(npx))
        return ext, ang, npx

    def lbd(self, dx, dy, ext, ang, npx):
        dx = np.ascontiguousarray(dx, dtype=np.int16)
        dy = np.ascontiguousarray(dy, dtype=np.int16)
        ext = np.ascontiguousarray(ext, dtype=np.float32)
        ang = np.ascontiguousarray(ang, dtype=np.float32)
        npx = np.ascontiguousarray(npx, dtype=np.int32)
        n = ext.shape[0]
        desc = np.empty((n, 72), np.float32)
        code = np.empty((n, 32), np.uint8)
        self.lib.lfo_lbd(_p(dx), _p(dy), dx.shape[0], dx.shape[1], _p(ext), _p(ang), _p(npx), n,
                         _p(desc), _p(code))
        return desc, code

    def match(self, q, t):
        q = np.ascontiguousarray(q, dtype=np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, dtype=np.uint8).reshape(-1, 32)
        idx = np.empty(q.shape[0], np.int32)
        dist = np.empty(q.shape[0], np.float32)
        self.lib.lfo_match(_p(q), q.shape[0], _p(t), t.shape[0], _p(idx), _p(dist))
        return idx, dist

    def match_mih(self, q, t):
        """Nearest neighbour with the reference's tie rule (first discovered by Mihasher::query,
        binary_descriptor_matcher.cpp:635-753): (idx, dist, number of equally near train codes)."""
        q = np.ascontiguousarray(q, dtype=np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, dtype=np.uint8).reshape(-1, 32)
        idx = np.empty(q.shape[0], np.int32)
        dist = np.empty(q.shape[0], np.float32)
        ties = np.empty(q.shape[0], np.int32)
        self.lib.lfo_match_mih(_p(q), q.shape[0], _p(t), t.shape[0], _p(idx), _p(dist), _p(ties))
        return idx, dist, ties

    def knn_match(self, q, t, k, tie_rule="lowest"):
        q = np.ascontiguousarray(q, dtype=np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, dtype=np.uint8).reshape(-1, 32)
        idx = np.empty((q.shape[0], k), np.int32)
        dist = np.empty((q.shape[0], k), np.float32)
        getattr(self.lib, "lfo_knn_match_mih" if tie_rule == "mihasher" else "lfo_knn_match")(_p(q), q.shape[0], _p(t), t.shape[0], int(k), _p(idx), _p(dist))
        return idx, dist

    def radius_match(self, q, t, max_distance, tie_rule="lowest"):
        q = np.ascontiguousarray(q, dtype=np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, dtype=np.uint8).reshape(-1, 32)
        fn = getattr(self.lib, "lfo_radius_match_mih" if tie_rule == "mihasher" else "lfo_radius_match")
        fn.restype = ctypes.c_int
        offsets = np.zeros(q.shape[0] + 1, np.int32)
        total = fn(_p(q), q.shape[0], _p(t), t.shape[0], ctypes.c_float(max_distance), _p(offsets), None, None)
        idx, dist = np.empty(max(total, 1), np.int32), np.empty(max(total, 1), np.float32)
        fn(_p(q), q.shape[0], _p(t), t.shape[0], ctypes.c_float(max_distance), _p(offsets), _p(idx), _p(dist))
        return offsets, idx[:total], dist[:total]

    def set_width_of_band(self, w):
        """BinaryDescriptor::setWidthOfBand (binary_descriptor_custom.cpp:134-176) for every later descriptor of this PROCESS's oracle
        library (frames, KeyLines, compute): 7 by default."""
        self.lib.lfo_lbd_set_width_of_band(int(w))

    def set_lsd_seed_order(self, mode):
        """0: raster inside a gradient bin (default); 1: libstdc++ std::sort order (the later OpenCV 3.x)."""
        self.lib.lfo_lsd_set_seed_order(int(mode))

    def match_float(self, q, t):
        q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1, 72)
        t = np.ascontiguousarray(t, dtype=np.float32).reshape(-1, 72)
        idx = np.empty(q.shape[0], np.int32)
        dist = np.empty(q.shape[0], np.float32)
        self.lib.lfo_match_float(_p(q), q.shape[0], _p(t), t.shape[0], _p(idx), _p(dist))
        return idx, dist

    # ---- whole frame
    def process_frame(self, bgr_in, cap=4096, describe=True):
        bgr_in = np.ascontiguousarray(bgr_in, dtype=np.uint8)
        assert bgr_in.shape == (self.c.in_rows, self.c.in_cols, 3), bgr_in.shape
        arr = {
            "lines": np.zeros((cap, 4), np.float32), "normals": np.zeros((cap, 2), np.float32),
            "color": np.zeros(cap, np.uint8), "pixels_normalized": np.zeros((cap, 4), np.float32),
            "ground": np.zeros((cap, 4), np.float64), "keep": np.zeros(cap, np.uint8),
            "desc": np.zeros((cap, 72), np.float32), "code": np.zeros((cap, 32), np.uint8),
        }
        fo = _FrameOut()
        for k, v in arr.items():
            setattr(fo, k, v.ctypes.data)
        if not describe:
            fo.desc = None
            fo.code = None
        n = self.lib.lfo_process_frame(ctypes.byref(self.c), _p(bgr_in), ctypes.byref(fo), cap)
        out = {k: v[:n].copy() for k, v in arr.items()}
        out["n"] = n
        out["n_color"] = [fo.n_color[i] for i in range(3)]
        return out


def _process_frame_edlines(self, bgr_in, params=None, describe=True):
    """The oracle's pieces composed for the EDLines detector of the batched path (include/lanefront.h, lf_set_detector;
    the contract of lf_set_image_edlines, frame by frame): EDLines (one octave) on BGR2GRAY of the working image, a line
    to every colour whose dilated mask is set under its truncated, clamped centre, then the reference's _findNormal /
    ordering, normalisation, ground projection, line sanity and BinaryDescriptor::compute's LBD -- same dict as
    process_frame.  None when the detector gives up on the frame."""
    work = self.preprocess(bgr_in)
    gray = self.bgr2gray(work)
    k = octave_keylines(gray, 1, params)
    if k is None:
        return None
    bw = self.color_masks(self.bgr2hsv(work))
    io = k["in_octave"]
    cx = ((io[:, 0] + io[:, 2]) / np.float32(2)).astype(np.int64).clip(0, gray.shape[1] - 1)
    cy = ((io[:, 1] + io[:, 3]) / np.float32(2)).astype(np.int64).clip(0, gray.shape[0] - 1)
    lines, normals, color = [], [], []
    for ci in range(3):
        area = self.dilate(bw[ci])
        sel = io[area[cy, cx] > 0]
        if len(sel) == 0:
            continue
        ol, on, _ = self.find_normals(area, sel.copy())
        lines.append(ol)
        normals.append(on.astype(np.float32))
        color.append(np.full(len(ol), ci, np.uint8))
    n = sum(len(a) for a in lines)
    r = {"n": n}
    r["lines"] = np.concatenate(lines) if n else np.zeros((0, 4), np.float32)
    r["normals"] = np.concatenate(normals) if n else np.zeros((0, 2), np.float32)
    r["color"] = np.concatenate(color) if n else np.zeros(0, np.uint8)
    r["pixels_normalized"] = self.normalize_lines(r["lines"]) if n else np.zeros((0, 4), np.float32)
    r["ground"] = self.ground_project(r["pixels_normalized"]) if n else np.zeros((0, 4), np.float64)
    r["keep"] = self.line_sanity(r["ground"], r["color"])[0] if n else np.zeros(0, np.uint8)
    if describe and n:
        dx, dy = self.sobel3(self.gaussian5(gray))
        ext, ang, npx = self.keylines(r["lines"], gray.shape[0], gray.shape[1])
        r["desc"], r["code"] = self.lbd(dx, dy, ext, ang, npx)
    else:
        r["desc"], r["code"] = np.zeros((0, 72), np.float32), np.zeros((0, 32), np.uint8)
    return r


Oracle.process_frame_edlines = _process_frame_edlines


def lsd_octave_keylines(gray, n_octaves=1, describe=True, seed_order="opencv30", options=None, mask=None):
    """LSDDetectorC::detect (ref: src/line_descriptor/src/LSDDetector_custom.cpp:49-72, 130-215) + BinaryDescriptor::compute,
    composed from the oracle's pieces: pyrDown pyramid (no blur), cv LSD with createLineSegmentDetector()'s defaults on every level,
    KeyLine fill (:164-197), descriptors on compute's own pyramid.  Same dict keys as octave_keylines.
    options: dict of LSDOptions fields (:218-325 detect / :327-438 detectFast, the same text twice): the detector's parameters and
    `length > min_length` (:277), class_id counting the kept lines (:293); mask: u8 image -- KeyLines whose two end points both lie on
    zero pixels are erased, correctly (keyCounter-- after the erase, :203-213 / :312-322)."""
    from lane_slam_amd import default_config
    gray = np.ascontiguousarray(gray, np.uint8)
    out = {k: [] for k in ("start_end", "in_octave", "angle", "num_pixels", "line_length", "octave", "class_id", "response", "size", "pt")}
    level = gray
    cls = 0
    for o in range(n_octaves):
        rows, cols = level.shape
        cfg = default_config("parity")
        cfg["in_size"] = [rows, cols]; cfg["img_size"] = [rows, cols]; cfg["top_cutoff"] = 0
        cfg["lsd"] = {"refine": 1, "scale": 0.8, "sigma_scale": 0.6, "quant": 2.0, "ang_th": 22.5, "log_eps": 0.0, "density_th": 0.7,
                      "n_bins": 1024, "seed_order": seed_order}
        if options:
            cfg["lsd"].update({k: v for k, v in options.items() if k != "min_length"})
        oc = Oracle(cfg)
        lines = np.asarray(oc.lsd(level, cap=20000), np.float32).reshape(-1, 4)
        ext, _, npx = oc.keylines(lines, rows, cols)
        if options is not None:                                # (:277) length > opts.min_length, the float-rounded length against the double
            ddx = (ext[:, 0] - ext[:, 2]).astype(np.float32).astype(np.float64)
            ddy = (ext[:, 1] - ext[:, 3]).astype(np.float32).astype(np.float64)
            kept = np.sqrt(ddx * ddx + ddy * ddy).astype(np.float32).astype(np.float64) > float(options.get("min_length", 0.0))
            ext, npx = ext[kept], npx[kept]
        n = ext.shape[0]
        scale = np.float32(1 << o)
        se = (ext * scale).astype(np.float32)
        dx = (ext[:, 0] - ext[:, 2]).astype(np.float32).astype(np.float64)
        dy = (ext[:, 1] - ext[:, 3]).astype(np.float32).astype(np.float64)
        length = np.sqrt(dx * dx + dy * dy).astype(np.float32)
        lib = ctypes.CDLL(_SO)
        lib.lfo_atan2.restype = ctypes.c_double
        lib.lfo_atan2.argtypes = [ctypes.c_double, ctypes.c_double]
        ang = np.array([np.float32(lib.lfo_atan2(float(np.float32(se[i, 3] - se[i, 1])), float(np.float32(se[i, 2] - se[i, 0])))) for i in range(n)], np.float32)
        out["start_end"].append(se); out["in_octave"].append(ext); out["angle"].append(ang); out["num_pixels"].append(npx)
        out["line_length"].append(length); out["octave"].append(np.full(n, o, np.int32)); out["class_id"].append(np.arange(cls, cls + n, dtype=np.int32))
        out["response"].append((length / np.float32(max(cols, rows))).astype(np.float32))
        out["size"].append(((se[:, 2] - se[:, 0]) * (se[:, 3] - se[:, 1])).astype(np.float32))
        out["pt"].append(np.stack([(se[:, 2] + se[:, 0]) / np.float32(2), (se[:, 3] + se[:, 1]) / np.float32(2)], axis=1).astype(np.float32))
        cls += n
        if o + 1 < n_octaves:
            level = pyrdown_u8(level)
    r = {k: (np.concatenate(v) if v else np.zeros(0)) for k, v in out.items()}
    if mask is not None and cls:
        se = r["start_end"]
        sx, sy, ex, ey = (se[:, i].astype(np.int64) for i in range(4))     # (int) of a non-negative float: truncation
        keep = ~((mask[sy, sx] == 0) & (mask[ey, ex] == 0))
        r = {k: v[keep] for k, v in r.items()}
        cls = int(keep.sum())
    r["n"] = cls
    if describe and cls:
        r["desc"], r["code"] = describe_keylines(gray, r["in_octave"], r["angle"], r["num_pixels"], r["octave"])
    else:
        r["desc"], r["code"] = np.zeros((0, 72), np.float32), np.zeros((0, 32), np.uint8)
    return r


def _jpeg_lib():
    build()
    lib = ctypes.CDLL(_SO)
    lib.lfo_jpeg_info.restype = ctypes.c_int
    lib.lfo_jpeg_decode.restype = ctypes.c_int
    return lib


def jpeg_info(data):
    """(rows, cols, components, hmax, vmax) of a JPEG stream; raises ValueError like image_cv_from_jpg."""
    lib = _jpeg_lib()
    buf = np.frombuffer(bytes(data), np.uint8)
    v = [ctypes.c_int() for _ in range(5)]
    rc = lib.lfo_jpeg_info(_p(buf), ctypes.c_size_t(buf.size), *[ctypes.byref(x) for x in v])
    if rc != 0:
        raise ValueError("Could not decode image (oracle rc %d)" % rc)
    return tuple(x.value for x in v)


def jpeg_decode(data):
    """BGR u8 image cv2.imdecode(data, IMREAD_COLOR) would return (oracle/lf_oracle_jpeg.c)."""
    lib = _jpeg_lib()
    rows, cols = jpeg_info(data)[:2]
    buf = np.frombuffer(bytes(data), np.uint8)
    out = np.empty((rows, cols, 3), np.uint8)
    rc = lib.lfo_jpeg_decode(_p(buf), ctypes.c_size_t(buf.size), _p(out))
    if rc != 0:
        raise ValueError("Could not decode image (oracle rc %d)" % rc)
    return out


def jpeg_coefficients(data):
    """Quantised coefficients int16 [blocks in scan order][64 natural order] (host-decoder check)."""
    lib = _jpeg_lib()
    rows, cols, ncomp, hmax, vmax = jpeg_info(data)
    mcux = (cols + 8 * hmax - 1) // (8 * hmax)
    mcuy = (rows + 8 * vmax - 1) // (8 * vmax)
    nblocks = mcux * mcuy * (1 if ncomp == 1 else hmax * vmax + 2)
    buf = np.frombuffer(bytes(data), np.uint8)
    out = np.zeros((nblocks, 64), np.int16)
    n = ctypes.c_int()
    lib.lfo_jpeg_coefficients.restype = ctypes.c_int
    rc = lib.lfo_jpeg_coefficients(_p(buf), ctypes.c_size_t(buf.size), _p(out), nblocks, ctypes.byref(n))
    if rc != 0 or n.value != nblocks:
        raise ValueError("Could not decode image (oracle rc %d)" % rc)
    return out


def detmath_lib():
    build()
    return ctypes.CDLL(_SO)


class LfoMapConfig(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int32) for k in ("capacity", "color_gating", "max_distance", "policy", "kept_only",
                                              "merge_distance", "when_full")]


class OracleMap(object):
    """Sequential statement of the live map + associator contract (oracle/lf_oracle_map.c).  Same keyword arguments as
    lane_slam_amd.LineAssociator."""

    def __init__(self, capacity=65536, color_gating=False, max_distance=128, policy="append", kept_only=True,
                 merge_distance=0, when_full="ring", tie_rule="mihasher"):
        build()
        self.lib = ctypes.CDLL(_SO)
        self.lib.lfo_map_create.restype = ctypes.c_void_p
        for f in ("lfo_map_codes", "lfo_map_colors", "lfo_map_ground", "lfo_map_hits", "lfo_map_last_seen"):
            getattr(self.lib, f).restype = ctypes.c_void_p
            getattr(self.lib, f).argtypes = [ctypes.c_void_p]
        c = LfoMapConfig(int(capacity), int(bool(color_gating)), int(max_distance), {"append": 0, "merge": 1}[policy],
                         int(bool(kept_only)), int(merge_distance), {"ring": 0, "error": 1}[when_full])
        self.capacity = int(capacity)
        self.m = ctypes.c_void_p(self.lib.lfo_map_create(ctypes.byref(c)))
        self.lib.lfo_map_set_tie_rule.argtypes = [ctypes.c_void_p, ctypes.c_int]
        self.lib.lfo_map_set_tie_rule(self.m, {"lowest": 0, "mihasher": 1}[tie_rule])

    def __del__(self):
        try:
            if self.m:
                self.lib.lfo_map_destroy(self.m)
                self.m = None
        except Exception:
            pass

    def state(self):
        size, head, ovf = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        ta, tr = ctypes.c_longlong(), ctypes.c_longlong()
        self.lib.lfo_map_state(self.m, ctypes.byref(size), ctypes.byref(head), ctypes.byref(ovf), ctypes.byref(ta), ctypes.byref(tr))
        return {"size": size.value, "head": head.value, "overflow": ovf.value, "total_appended": ta.value,
                "total_refreshed": tr.value}

    def seed(self, codes, colors=None, ground=None):
        codes = np.ascontiguousarray(codes, np.uint8).reshape(-1, 32)
        colors = None if colors is None else np.ascontiguousarray(colors, np.uint8)
        ground = None if ground is None else np.ascontiguousarray(ground, np.float64).reshape(-1, 4)
        self.lib.lfo_map_seed(self.m, _p(codes), None if colors is None else _p(colors),
                              None if ground is None else _p(ground), codes.shape[0])

    def associate(self, codes, colors=None):
        codes = np.ascontiguousarray(codes, np.uint8).reshape(-1, 32)
        colors = None if colors is None else np.ascontiguousarray(colors, np.uint8)
        n = codes.shape[0]
        idx, dist = np.empty(n, np.int32), np.empty(n, np.float32)
        self.lib.lfo_map_associate(self.m, _p(codes), None if colors is None else _p(colors), n, _p(idx), _p(dist))
        return idx, dist

    def to_map_frame(self, ground, frame_offset, poses):
        ground = np.ascontiguousarray(ground, np.float64).reshape(-1, 4)
        fo = np.ascontiguousarray(frame_offset, np.int32)
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 3)
        out = ground.copy()
        self.lib.lfo_map_to_map_frame(_p(ground), ground.shape[0], _p(fo), poses.shape[0], _p(poses), _p(out))
        return out

    def update(self, codes, colors, keep, ground, idx, dist, step):
        codes = np.ascontiguousarray(codes, np.uint8).reshape(-1, 32)
        n = codes.shape[0]
        colors = None if colors is None else np.ascontiguousarray(colors, np.uint8)
        keep = None if keep is None else np.ascontiguousarray(keep, np.uint8)
        ground = None if ground is None else np.ascontiguousarray(ground, np.float64).reshape(-1, 4)
        idx = np.ascontiguousarray(idx, np.int32)
        dist = np.ascontiguousarray(dist, np.float32)
        self.lib.lfo_map_update(self.m, _p(codes), None if colors is None else _p(colors), None if keep is None else _p(keep),
                                None if ground is None else _p(ground), _p(idx), _p(dist), n, int(step))

    def step(self, codes, colors, keep, ground, step, frame_offset=None, poses=None):
        """associate + (pose transform) + update; returns (idx, dist)."""
        idx, dist = self.associate(codes, colors)
        g = ground
        if poses is not None and ground is not None:
            g = self.to_map_frame(ground, frame_offset, poses)
        self.update(codes, colors, keep, g, idx, dist, step)
        return idx, dist

    def fetch(self):
        n = self.capacity

        def arr(fn, dt, cols):
            p = getattr(self.lib, fn)(self.m)
            a = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(dt)), shape=(n * cols,))
            return a.reshape((n, cols) if cols > 1 else (n,)).copy()
        return {"code": arr("lfo_map_codes", ctypes.c_uint8, 32), "color": arr("lfo_map_colors", ctypes.c_uint8, 1),
                "ground": arr("lfo_map_ground", ctypes.c_double, 4), "hits": arr("lfo_map_hits", ctypes.c_int32, 1),
                "last_seen": arr("lfo_map_last_seen", ctypes.c_int32, 1)}


def kmeans(bgr_points, init, max_iter=25, tol=1e-4):
    """scikit-learn's Lloyd iteration from an explicit init, as kmeans.py:24-26 runs it.  bgr_points: [N, 3] u8.
    Returns (centers [k, 3] f64, counts [k] i64, inertia, n_iter)."""
    lib = ctypes.CDLL(build())
    lib.lfo_kmeans.restype = ctypes.c_int
    pts = np.ascontiguousarray(bgr_points, np.uint8).reshape(-1, 3)
    init = np.ascontiguousarray(init, np.float64).reshape(-1, 3)
    k = init.shape[0]
    centers = np.zeros((k, 3), np.float64)
    counts = np.zeros(k, np.int64)
    inertia = ctypes.c_double()
    it = lib.lfo_kmeans(_p(pts), pts.shape[0], k, _p(init), int(max_iter), ctypes.c_double(tol), _p(centers), _p(counts), ctypes.byref(inertia))
    if it < 0:
        raise ValueError("k-means: a cluster ran empty")
    return centers, counts, inertia.value, it


# ---------------------------------------------------------------- EDLines + multi-octave KeyLines (lf_oracle_edlines.c)
class LfoEdlinesParams(ctypes.Structure):
    _fields_ = [("ksize", ctypes.c_int32), ("sigma", ctypes.c_float), ("gradient_threshold", ctypes.c_int32),
                ("anchor_threshold", ctypes.c_int32), ("scan_intervals", ctypes.c_int32), ("min_line_len", ctypes.c_int32),
                ("line_fit_err_threshold", ctypes.c_double)]


class _KeylinesOut(ctypes.Structure):
    _fields_ = [(k, ctypes.c_void_p) for k in ("start_end", "in_octave", "angle", "num_pixels", "line_length", "octave",
                                               "class_id", "response", "size", "pt", "salience", "desc", "code")] + \
               [("octave_rows", ctypes.c_int32 * 8), ("octave_cols", ctypes.c_int32 * 8), ("octave_lines", ctypes.c_int32 * 8)]


def _edlib():
    lib = ctypes.CDLL(build())
    lib.lfo_edlines_run.restype = ctypes.c_void_p
    lib.lfo_edlines_array.restype = ctypes.c_void_p
    lib.lfo_ed_nfa.restype = ctypes.c_double
    return lib


def edlines_params(**kw):
    """EDLineDetector's defaults (binary_descriptor_custom.cpp:1374-1385), optionally overridden."""
    p = LfoEdlinesParams()
    _edlib().lfo_edlines_params_default(ctypes.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def gaussian_taps_q8(ksize, sigma):
    taps = np.zeros(ksize, np.int32)
    _edlib().lfo_gaussian_taps_q8(int(ksize), ctypes.c_double(sigma), _p(taps))
    return taps


def gaussian_blur_u8(img, ksize, sigma):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty_like(img)
    _edlib().lfo_gaussian_blur_u8(_p(img), img.shape[0], img.shape[1], int(ksize), ctypes.c_double(sigma), _p(out))
    return out


def resize_linear_u8(img, inv_scale):
    img = np.ascontiguousarray(img, np.uint8)
    lib = _edlib()
    r, c = ctypes.c_int(), ctypes.c_int()
    lib.lfo_resize_size(img.shape[0], img.shape[1], ctypes.c_double(inv_scale), ctypes.byref(r), ctypes.byref(c))
    out = np.empty((r.value, c.value), np.uint8)
    lib.lfo_resize_linear_u8(_p(img), img.shape[0], img.shape[1], ctypes.c_double(inv_scale), _p(out))
    return out


def pyrdown_u8(img):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty((img.shape[0] // 2, img.shape[1] // 2), np.uint8)
    _edlib().lfo_pyrdown_u8(_p(img), img.shape[0], img.shape[1], _p(out))
    return out


def ed_nfa(n, k, p, log_nt):
    return _edlib().lfo_ed_nfa(int(n), int(k), ctypes.c_double(p), ctypes.c_double(log_nt))


_ED_ARRAYS = [("dx", np.int16), ("dy", np.int16), ("g", np.int16), ("gwo", np.int16), ("dir", np.uint8), ("edge", np.uint8),
              ("ax", np.uint32), ("ay", np.uint32), ("xcors", np.uint32), ("ycors", np.uint32), ("sid", np.uint32),
              ("lx", np.uint32), ("ly", np.uint32), ("lsid", np.uint32), ("equations", np.float64), ("endpoints", np.float32),
              ("direction", np.float32), ("salience", np.float32)]


def edlines(blurred, params=None):
    """EDLineDetector::EDline on an already blurred u8 image, every stage kept.  None when the detector gives up."""
    lib = _edlib()
    img = np.ascontiguousarray(blurred, np.uint8)
    rows, cols = img.shape
    params = params or edlines_params()
    e = lib.lfo_edlines_run(ctypes.byref(params), _p(img), rows, cols)
    if not e:
        return None
    e = ctypes.c_void_p(e)
    cnt = [ctypes.c_int() for _ in range(5)]
    lib.lfo_edlines_counts(e, *[ctypes.byref(c) for c in cnt])
    n_anchors, n_edges, n_edge_px, n_lines, n_line_px = (c.value for c in cnt)
    sizes = {"dx": rows * cols, "dy": rows * cols, "g": rows * cols, "gwo": rows * cols, "dir": rows * cols, "edge": rows * cols,
             "ax": n_anchors, "ay": n_anchors, "xcors": n_edge_px, "ycors": n_edge_px, "sid": n_edges + 1, "lx": n_line_px,
             "ly": n_line_px, "lsid": n_lines + 1, "equations": 3 * n_lines, "endpoints": 4 * n_lines, "direction": n_lines,
             "salience": n_lines}
    out = {"n_anchors": n_anchors, "n_edges": n_edges, "n_lines": n_lines}
    for which, (name, dt) in enumerate(_ED_ARRAYS):
        n = sizes[name]
        ptr = lib.lfo_edlines_array(e, which)
        a = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(max(n, 1),))[:n].copy()
        out[name] = a
    for k in ("dx", "dy", "g", "gwo", "dir", "edge"):
        out[k] = out[k].reshape(rows, cols)
    out["equations"] = out["equations"].reshape(-1, 3)
    out["endpoints"] = out["endpoints"].reshape(-1, 4)
    lib.lfo_edlines_free(e)
    return out


def erase_by_mask_as_written(start_end, mask):
    """BinaryDescriptor::detectImpl's mask loop AS WRITTEN (ref: binary_descriptor_custom.cpp:509-519):
        for (keyCounter = 0; keyCounter < keylines.size(); keyCounter++)
            if (mask(start) == 0 && mask(end) == 0) keylines.erase(keylines.begin() + keyCounter);
    -- no step back after the erase, so the element that slides into the erased place is skipped.  Returns the indices (into
    start_end) of the KeyLines that remain."""
    idx = list(range(len(start_end)))
    k = 0
    while k < len(idx):
        sx, sy, ex, ey = (int(v) for v in start_end[idx[k]])
        if mask[sy, sx] == 0 and mask[ey, ex] == 0:
            del idx[k]
        k += 1
    return np.asarray(idx, np.int64)


def octave_keylines(gray, n_octaves=1, params=None, ksize=5, cap=20000):
    """BinaryDescriptor::operator() (detect with EDLines over n_octaves + LBD on the detector's gradients): dict of
    per-KeyLine arrays in detectImpl's order, plus 'octave_size' [(rows, cols)] and 'octave_lines'."""
    lib = _edlib()
    lib.lfo_octave_keylines.restype = ctypes.c_int
    gray = np.ascontiguousarray(gray, np.uint8)
    params = params or edlines_params()
    shapes = {"start_end": (np.float32, 4), "in_octave": (np.float32, 4), "angle": (np.float32, 1), "num_pixels": (np.int32, 1),
              "line_length": (np.float32, 1), "octave": (np.int32, 1), "class_id": (np.int32, 1), "response": (np.float32, 1),
              "size": (np.float32, 1), "pt": (np.float32, 2), "salience": (np.float32, 1), "desc": (np.float32, 72), "code": (np.uint8, 32)}
    arrs = {k: np.zeros((cap, c) if c > 1 else cap, dt) for k, (dt, c) in shapes.items()}
    o = _KeylinesOut()
    for k, a in arrs.items():
        setattr(o, k, a.ctypes.data)
    n = lib.lfo_octave_keylines(ctypes.byref(params), _p(gray), gray.shape[0], gray.shape[1], int(n_octaves), int(ksize), int(cap), ctypes.byref(o))
    if n < 0:
        return None
    out = {k: a[:n].copy() for k, a in arrs.items()}
    out["n"] = n
    out["octave_size"] = [(o.octave_rows[i], o.octave_cols[i]) for i in range(n_octaves)]
    out["octave_lines"] = [o.octave_lines[i] for i in range(n_octaves)]
    return out


def describe_keylines(gray, in_octave, angle, num_pixels, octave):
    """BinaryDescriptor::compute on given KeyLines (pyramid of computeGaussianPyramid): (desc [n, 72], code [n, 32])."""
    lib = _edlib()
    lib.lfo_describe_keylines.restype = ctypes.c_int
    gray = np.ascontiguousarray(gray, np.uint8)
    io = np.ascontiguousarray(in_octave, np.float32).reshape(-1, 4)
    n = io.shape[0]
    ang = np.ascontiguousarray(angle, np.float32)
    npx = np.ascontiguousarray(num_pixels, np.int32)
    octv = np.ascontiguousarray(octave, np.int32)
    desc, code = np.zeros((n, 72), np.float32), np.zeros((n, 32), np.uint8)
    rc = lib.lfo_describe_keylines(_p(gray), gray.shape[0], gray.shape[1], _p(io), _p(ang), _p(npx), _p(octv), n, _p(desc), _p(code))
    if rc < 0:
        raise ValueError("describe_keylines: octave out of range")
    return desc, code
