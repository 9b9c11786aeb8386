#!/usr/bin/env python3
"""Generate golden vectors from the pieces of the reference that RUN in the build container.

Run here only (needs /root/reference):   python3 tests/golden/make_golden.py
Outputs small .npz fixtures next to this script; nothing of the reference's source is
stored, only numeric inputs and the outputs its own code produced for them.

What is imported from /root/reference (Python 3 can execute these files as they are):
  * src/line_detector/include/line_detector/line_detector_lsd.py
        LineDetectorLSD._findNormal / _checkBounds / _correctPixelOrdering   (a-5)
  * src/line_sanity/src/line_sanity_node.py
        LineSanityNode.processSegmentList / fancyFilters                     (a-8)
  * src/anti_instagram/include/anti_instagram/scale_and_shift.py
        scaleandshift2                                                       (a-1)
  * src/ground_projection/include/ground_projection/GroundProjection.py (Python 2: converted in memory by lib2to3)
        GroundProjection.vector2pixel / pixel2ground with rectifyPoint = identity   (a-7 minus undistortPoints)
  * src/line_detector/src/line_detector_node.py
        LineDetectorNode.processImage_ / toSegmentMsg with a FAKE detector plugin that returns fixed lines and
        normals: the normalisation arithmetic, the colour order, the empty-list rule               (a-6)
  * src/ground_projection/src/ground_projection_node.py
        GroundProjectionNode.lineseglist_cb: which fields travel on, in which order                 (a-7 glue)
  * src/duckietown_msgs/msg/{SegmentList,Segment,Vector2D}.msg
        parsed here to serialise the messages the two nodes built -> the ROS 1 wire bytes           (f-2)
  * src/anti_instagram/include/anti_instagram/kmeans.py
        runKMeans / getimgdatapts with the scikit-learn installed in this image (1.7.2: the reference does not pin
        a version): cluster centres, label counts and score for the reference's two inits              (f-4, k-means part)
The modules `rospy`, `cv2`, `*_msgs.msg` they import at file scope are absent from this
image; they are replaced by empty name-only stubs (no arithmetic) so the import
statements succeed.  Message classes are plain attribute holders with the constants of
src/duckietown_msgs/msg/Segment.msg:1-3.
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference/src"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Obj(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)


class Segment(object):
    WHITE, YELLOW, RED = 0, 1, 2

    def __init__(self):
        self.color = 0
        self.pixels_normalized = [_Obj(x=0.0, y=0.0), _Obj(x=0.0, y=0.0)]
        self.normal = _Obj(x=0.0, y=0.0)
        self.points = [_Obj(x=0.0, y=0.0, z=0.0), _Obj(x=0.0, y=0.0, z=0.0)]


class _Stamp(object):
    def __init__(self, secs=0, nsecs=0):
        self.secs, self.nsecs = secs, nsecs

    def to_sec(self):
        return self.secs + 1e-9 * self.nsecs


class SegmentList(object):
    def __init__(self):
        self.header = _Obj(seq=0, stamp=_Stamp(), frame_id="")
        self.segments = []


def install_stubs():
    class _Pub(object):
        def __init__(self, *a, **k):
            pass

        def publish(self, *a):
            pass

    _stub("rospy", get_param=lambda *a, **k: "veh", Subscriber=_Pub, Publisher=_Pub,
          loginfo=lambda *a, **k: None, init_node=lambda *a, **k: None,
          on_shutdown=lambda *a, **k: None, spin=lambda: None)
    _stub("cv2")
    _stub("duckietown_msgs")
    _stub("duckietown_msgs.msg", AntiInstagramTransform=_Obj, BoolStamped=_Obj, Segment=Segment,
          SegmentList=SegmentList, Vector2D=_Obj)
    _stub("geometry_msgs")
    _stub("geometry_msgs.msg", Point=_Obj)
    _stub("sensor_msgs")
    _stub("sensor_msgs.msg", CompressedImage=_Obj, Image=_Obj)
    _stub("visualization_msgs")
    _stub("visualization_msgs.msg", Marker=_Obj)


def load_file(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def golden_find_normal():
    # duckietown_utils/__init__ needs frozendict/ruamel: load parameters.py directly
    pkg = _stub("duckietown_utils")
    pkg.__path__ = []
    load_file("duckietown_utils.parameters", REF + "/duckietown/include/duckietown_utils/parameters.py")
    sys.path.insert(0, REF + "/line_detector/include")
    from line_detector.line_detector_lsd import LineDetectorLSD
    conf = {k: 0 for k in ["hsv_white1", "hsv_white2", "hsv_yellow1", "hsv_yellow2", "hsv_red1", "hsv_red2",
                           "hsv_red3", "hsv_red4", "dilation_kernel_size", "canny_thresholds",
                           "hough_threshold", "hough_min_line_length", "hough_max_line_gap"]}
    det = LineDetectorLSD(conf)
    rng = np.random.default_rng(20261002)
    cases = {}
    for ci, (rows, cols, n) in enumerate([(80, 160, 64), (320, 640, 200), (7, 9, 40)]):
        bw = (rng.random((rows, cols)) < 0.5).astype(np.uint8) * 255
        # blobs so that both sign outcomes occur with structure, not only noise
        bw[rows // 4: rows // 2, cols // 4: cols // 2] = 255
        bw[rows // 2:, : cols // 3] = 0
        lines = np.empty((n, 4), np.float32)
        lines[:, 0::2] = rng.uniform(-4, cols + 4, (n, 2)).astype(np.float32)   # some out of bounds
        lines[:, 1::2] = rng.uniform(-4, rows + 4, (n, 2)).astype(np.float32)
        lines[0] = [1.5, 1.5, 1.5, 6.25]           # vertical
        lines[1] = [2.0, 3.0, 8.0, 3.0]            # horizontal
        lines[2] = [cols - 1.0, rows - 1.0, cols - 3.0, rows - 2.5]
        lines_in = lines.copy()
        centers, normals = det._findNormal(bw, lines)   # mutates `lines` (reordering)
        cases["bw%d" % ci] = bw
        cases["lines_in%d" % ci] = lines_in
        cases["lines_out%d" % ci] = np.asarray(lines, np.float32)
        cases["normals%d" % ci] = np.asarray(normals, np.float64)
        cases["centers%d" % ci] = np.asarray(centers, np.float32)
    cases["n_cases"] = np.int32(3)
    np.savez_compressed(os.path.join(OUT, "find_normal.npz"), **cases)
    print("find_normal: dtypes", normals.dtype, centers.dtype, lines.dtype)


def golden_line_sanity():
    mod = load_file("ref_line_sanity_node", REF + "/line_sanity/src/line_sanity_node.py")
    node = mod.LineSanityNode()
    rng = np.random.default_rng(7)
    n = 600
    pts = np.empty((n, 4), np.float64)
    pts[:, 0] = rng.uniform(-0.2, 1.2, n)
    pts[:, 1] = rng.uniform(-0.5, 0.5, n)
    pts[:, 2] = pts[:, 0] + rng.uniform(-0.3, 0.3, n)
    pts[:, 3] = pts[:, 1] + rng.uniform(-0.3, 0.3, n)
    color = rng.integers(0, 3, n).astype(np.uint8)
    # edge cases: degenerate, behind, exact survey example, vertical, horizontal
    pts[0] = [0.3, 0.1, 0.3, 0.1]; color[0] = 0
    pts[1] = [0.3, 0.1, 0.3, 0.1]; color[1] = 1
    pts[2] = [0.30, -0.10, 0.20, -0.11]; color[2] = 0
    pts[3] = [-0.01, 0.0, 0.5, 0.0]; color[3] = 0
    pts[4] = [0.5, 0.0, 0.5, 0.2]; color[4] = 1
    pts[5] = [0.2, 0.05, 0.6, 0.05]; color[5] = 1
    pts[6] = [0.2, 0.05, 0.6, 0.05]; color[6] = 2
    sl = SegmentList()
    for i in range(n):
        s = Segment()
        s.color = int(color[i])
        s.points[0].x, s.points[0].y = float(pts[i, 0]), float(pts[i, 1])
        s.points[1].x, s.points[1].y = float(pts[i, 2]), float(pts[i, 3])
        s.idx = i
        sl.segments.append(s)
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):     # the reference prints on RED segments
        filt = node.processSegmentList(sl)
    keep = np.zeros(n, np.uint8)
    for s in filt.segments:
        keep[s.idx] = 1
    d = np.full(n, np.nan)
    phi = np.full(n, np.nan)
    l = np.full(n, np.nan)
    state = np.zeros(n, np.int32)
    with np.errstate(all="ignore"):
        for i, s in enumerate(sl.segments):
            d[i], phi[i], l[i], state[i] = node.fancyFilters(s)
    consts = np.array([node.lanewidth, node.linewidth_white, node.linewidth_yellow, node.d_min,
                       node.d_max, node.phi_min, node.phi_max], np.float64)
    np.savez_compressed(os.path.join(OUT, "line_sanity.npz"), pts=pts, color=color, keep=keep,
                        d=d, phi=phi, l=l, state=state, consts=consts)
    print("line_sanity: kept %d / %d" % (keep.sum(), n), "example", d[2], phi[2], state[2])


def golden_scaleandshift():
    pkg = _stub("anti_instagram", logger=None)
    pkg.__path__ = [REF + "/anti_instagram/include/anti_instagram"]
    sas = load_file("anti_instagram.scale_and_shift", REF + "/anti_instagram/include/anti_instagram/scale_and_shift.py")
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (24, 32, 3), dtype=np.uint8)
    img[0, :8, :] = np.array([[0, 1, 2], [253, 254, 255], [127, 128, 129], [5, 50, 200],
                              [255, 0, 255], [64, 64, 64], [1, 1, 1], [200, 100, 10]], np.uint8)
    scales = np.array([[1.0, 1.0, 1.0], [1.1732, 0.9421, 1.3007], [0.5, 2.0, -1.0], [1.7, 1.2, 0.3]], np.float64)
    shifts = np.array([[0.0, 0.0, 0.0], [-12.25, 7.5, 3.3301], [0.5, -0.5, 255.0], [-30.0, 20.49, 0.51]], np.float64)
    outs = np.stack([sas.scaleandshift2(img, list(scales[i]), list(shifts[i])) for i in range(len(scales))])
    assert outs.dtype == np.float32
    np.savez_compressed(os.path.join(OUT, "scaleandshift.npz"), img=img, scales=scales, shifts=shifts, out=outs)
    print("scaleandshift:", outs.shape, outs.dtype)


def golden_ground_projection():
    """GroundProjection.vector2pixel + pixel2ground (a-7 without the undistortion).  The file is Python 2
    (`print ob`), so it is converted IN MEMORY with lib2to3 and executed; nothing of it is stored.  The
    instance is built without __init__ (which needs ROS parameter plumbing): H comes from the reference's own
    default extrinsic calibration file, ci_ is a 640x480 CameraInfo holder, and pcm_.rectifyPoint -- which is
    cv2.undistortPoints inside the absent third-party image_geometry package -- is replaced by the identity.
    The vectors therefore pin the pixel scaling, the four clamps (including `v > ch-1 -> 0`), the always-on
    rectification call, the homography product and the division; NOT undistortPoints itself."""
    import yaml
    from lib2to3.refactor import RefactoringTool, get_fixers_from_package
    path = REF + "/ground_projection/include/ground_projection/GroundProjection.py"
    src3 = str(RefactoringTool(get_fixers_from_package("lib2to3.fixes")).refactor_string(open(path).read() + "\n", path))

    class _Pcm(object):
        def rectifyPoint(self, uv):
            return uv

    _stub("image_geometry", PinholeCameraModel=_Pcm)
    du = _stub("duckietown_utils", logger=None, get_duckiefleet_root=lambda: "")
    du.__path__ = []
    _stub("duckietown_utils.path_utils", get_ros_package_path=lambda *a: "")
    _stub("duckietown_utils.yaml_wrap", yaml_load_file=lambda *a: None, yaml_write_to_file=lambda *a: None)
    sys.modules["sensor_msgs.msg"].CameraInfo = _Obj
    sys.modules["duckietown_msgs.msg"].Pixel = _Obj
    mod = types.ModuleType("ref_ground_projection")
    exec(compile(src3, path, "exec"), mod.__dict__)
    gp = object.__new__(mod.GroundProjection)
    gp.rectified_input = False
    ext = yaml.safe_load(open(REF + "/duckietown/include/calibrations/camera_extrinsic/default.yaml"))
    gp.H = np.array(ext["homography"], np.float64).reshape(3, 3)
    gp.ci_ = _Obj(width=640, height=480)
    gp.pcm_ = _Pcm()
    rng = np.random.default_rng(31)
    n = 400
    vec = rng.uniform(-0.15, 1.15, (n, 2)).astype(np.float32)          # message fields are float32
    vec[0] = [0.0, 0.0]; vec[1] = [1.0, 1.0]; vec[2] = [0.5, 479.0 / 480.0]; vec[3] = [0.5, 479.5 / 480.0]
    vec[4] = [639.0 / 640.0, 0.25]; vec[5] = [639.5 / 640.0, 0.25]; vec[6] = [-0.001, -0.001]
    out = np.empty((n, 3), np.float64)
    pix = np.empty((n, 2), np.float64)
    for i in range(n):
        v = _Obj(x=float(vec[i, 0]), y=float(vec[i, 1]))
        px = gp.vector2pixel(v)
        pix[i] = [px.u, px.v]
        g = gp.vector2ground(v)
        out[i] = [g.x, g.y, g.z]
    np.savez_compressed(os.path.join(OUT, "ground_projection.npz"), vec=vec, pixel=pix, ground=out, H=gp.H,
                        cam=np.array([640, 480], np.int32))
    print("ground_projection:", n, "points; clamp quirk example", pix[3], out[3])


# ---------------------------------------------------------------------------------------------------------
# ROS 1 wire format from the reference's own .msg files.  std_msgs/Header and geometry_msgs/Point are not in the
# reference's tree (ROS packages); their public definitions are written out here.  Serialisation rules (public,
# wiki.ros.org/msg): little endian, fields in declaration order, constants skipped, fixed arrays back to back,
# variable arrays and strings behind a uint32 length, `time` = uint32 secs + uint32 nsecs.
_PUBLIC_DEFS = {
    "std_msgs/Header": "uint32 seq\ntime stamp\nstring frame_id\n",
    "geometry_msgs/Point": "float64 x\nfloat64 y\nfloat64 z\n",
}
_BUILTIN = {"uint8": "<B", "int8": "<b", "uint16": "<H", "int16": "<h", "uint32": "<I", "int32": "<i", "float32": "<f",
            "float64": "<d", "bool": "<B"}


def _msg_fields(type_name):
    if type_name in _PUBLIC_DEFS:
        text = _PUBLIC_DEFS[type_name]
    else:
        pkg, name = type_name.split("/")
        text = open(os.path.join(REF, pkg, "msg", name + ".msg")).read()
    fields = []
    for line in text.splitlines():
        line = line.split("#")[0].strip()
        if not line or "=" in line:
            continue
        t, n = line.split()
        fields.append((t, n))
    return fields


def ros_serialize(obj, type_name, pkg="duckietown_msgs"):
    import struct
    base, arr = type_name, None
    if type_name.endswith("]"):
        base, rest = type_name.split("[")
        arr = rest[:-1]
    if arr is not None:
        out = b"" if arr else struct.pack("<I", len(obj))
        if arr:
            assert len(obj) == int(arr)
        return out + b"".join(ros_serialize(v, base, pkg) for v in obj)
    if base in _BUILTIN:
        return struct.pack(_BUILTIN[base], obj)
    if base == "string":
        b = obj.encode()
        return struct.pack("<I", len(b)) + b
    if base == "time":
        return struct.pack("<II", obj.secs, obj.nsecs)
    if base == "Header":
        base = "std_msgs/Header"
    if "/" not in base:
        base = pkg + "/" + base
    return b"".join(ros_serialize(getattr(obj, n), t, base.split("/")[0]) for t, n in _msg_fields(base))


def golden_node_pipeline():
    """The reference's own node code around a-6 / a-7, executed here with a fake detector plugin:
    LineDetectorNode.processImage_ + toSegmentMsg (line_detector_node.py:141-213,251-265) and
    GroundProjectionNode.lineseglist_cb (ground_projection_node.py:55-65), and the messages they build serialised
    with the field order of the reference's .msg files.  cv2.convertScaleAbs / drawLines / cv_bridge are name-only
    stubs: the fake detector never looks at the image, so no arithmetic of theirs reaches the fixture."""
    import time
    import yaml
    from lib2to3.refactor import RefactoringTool, get_fixers_from_package
    if not hasattr(time, "clock"):
        time.clock = time.process_time                 # timekeeper.py:20 (Python 2 API), timing only
    rospy = sys.modules["rospy"]
    rospy.get_time = time.time
    sys.modules["cv2"].convertScaleAbs = lambda img: img
    _stub("cv_bridge", CvBridge=_Obj, CvBridgeError=Exception)
    ai = _stub("anti_instagram")
    ai.__path__ = []
    _stub("anti_instagram.AntiInstagram", AntiInstagram=_Obj)
    du = _stub("duckietown_utils", logger=None, get_duckiefleet_root=lambda: "")
    du.__path__ = []
    _stub("duckietown_utils.instantiate_utils", instantiate=lambda *a: None)
    _stub("duckietown_utils.jpg", image_cv_from_jpg=lambda data: data)       # the "decoded" image is handed over as is
    load_file("duckietown_utils.parameters", REF + "/duckietown/include/duckietown_utils/parameters.py")
    if REF + "/line_detector/include" not in sys.path:
        sys.path.insert(0, REF + "/line_detector/include")
    _stub("line_detector.line_detector_plot", color_segment=lambda *a: None, drawLines=lambda *a: None)
    from line_detector.line_detector_interface import Detections
    node_mod = load_file("ref_line_detector_node", REF + "/line_detector/src/line_detector_node.py")

    class FakeDetector(object):
        def __init__(self, per_color):
            self.per_color = per_color

        def setImage(self, bgr):
            self.shape = bgr.shape

        def detectLines(self, color):
            lines, normals = self.per_color[color]
            if len(lines) == 0:
                return Detections(lines=[], normals=[], area=None, centers=[])       # line_detector_lsd.py:68-71
            return Detections(lines=lines.copy(), normals=normals.copy(), area=None, centers=None)

    class Capture(object):
        def publish(self, msg):
            self.msg = msg

    # ground projection node around the converted GroundProjection class (see golden_ground_projection)
    path = REF + "/ground_projection/include/ground_projection/GroundProjection.py"
    src3 = str(RefactoringTool(get_fixers_from_package("lib2to3.fixes")).refactor_string(open(path).read() + "\n", path))

    class _Pcm(object):
        def rectifyPoint(self, uv):
            return uv

    _stub("image_geometry", PinholeCameraModel=_Pcm)
    _stub("duckietown_utils.path_utils", get_ros_package_path=lambda *a: "")
    _stub("duckietown_utils.yaml_wrap", yaml_load_file=lambda *a: None, yaml_write_to_file=lambda *a: None)
    sys.modules["sensor_msgs.msg"].CameraInfo = _Obj
    sys.modules["duckietown_msgs.msg"].Pixel = _Obj
    gpkg = _stub("ground_projection")
    gpkg.__path__ = []
    _stub("ground_projection.srv", **{k: _Obj for k in ("EstimateHomography", "EstimateHomographyResponse", "GetGroundCoord",
                                                        "GetGroundCoordResponse", "GetImageCoord", "GetImageCoordResponse")})
    gmod = types.ModuleType("ground_projection.GroundProjection")
    sys.modules["ground_projection.GroundProjection"] = gmod
    exec(compile(src3, path, "exec"), gmod.__dict__)
    gp = object.__new__(gmod.GroundProjection)
    gp.rectified_input = False
    ext = yaml.safe_load(open(REF + "/duckietown/include/calibrations/camera_extrinsic/default.yaml"))
    gp.H = np.array(ext["homography"], np.float64).reshape(3, 3)
    gp.ci_ = _Obj(width=640, height=480)
    gp.pcm_ = _Pcm()
    gnode_mod = load_file("ref_ground_projection_node", REF + "/ground_projection/src/ground_projection_node.py")
    gnode = object.__new__(gnode_mod.GroundProjectionNode)
    gnode.gp = gp
    gnode.pub_lineseglist_ = Capture()

    rng = np.random.default_rng(4242)
    cases = {}
    geoms = [((120, 160), 40, (37, 0, 5)), ((480, 640), 160, (60, 33, 0)), ((200, 300), 7, (9, 11, 4)), ((120, 160), 40, (0, 0, 0))]
    for ci, (size, cut, counts) in enumerate(geoms):
        H, W = size
        per = {}
        for color, n in zip(("white", "yellow", "red"), counts):
            lines = np.empty((n, 4), np.float32)
            lines[:, 0::2] = rng.uniform(-2, W + 2, (n, 2)).astype(np.float32)
            lines[:, 1::2] = rng.uniform(-2, H - cut + 2, (n, 2)).astype(np.float32)
            if n > 2:
                lines[0] = [0.0, 0.0, W - 1.0, H - cut - 1.0]
                lines[1] = [0.5, 1.25, 100.625, 33.3125]
            th = rng.uniform(-np.pi, np.pi, n)
            # LineDetectorLSD hands over float64 normals whose values are float32 products (line_detector_lsd.py:118-123)
            normals = np.column_stack([np.cos(th), np.sin(th)]).astype(np.float32).astype(np.float64)
            per[color] = (lines, normals)
        node = object.__new__(node_mod.LineDetectorNode)
        node.node_name = "LineDetectorNode"
        node.stats = node_mod.Stats()
        node.intermittent_interval, node.intermittent_counter = 100, 5
        node.image_size, node.top_cutoff = [H, W], cut
        node.ai = _Obj(applyTransform=lambda img: img)
        node.detector = FakeDetector(per)
        node.pub_lines, node.pub_image = Capture(), Capture()
        node.bridge = _Obj(cv2_to_imgmsg=lambda img, enc: _Obj(header=_Obj(stamp=None)))
        node.verbose = False
        img_msg = _Obj(data=np.zeros((H, W, 3), np.uint8), header=_Obj(stamp=_Stamp(1234, 5678)))
        node.processImage_(img_msg)
        sl = node.pub_lines.msg
        sl.header.seq, sl.header.frame_id = 7, "cam"
        n = len(sl.segments)
        assert n == sum(counts)
        cases["geom%d" % ci] = np.array([H, W, cut], np.int32)
        for color in ("white", "yellow", "red"):
            cases["lines_%s%d" % (color, ci)] = per[color][0]
            cases["normals_%s%d" % (color, ci)] = per[color][1]
        cases["det_color%d" % ci] = np.array([s.color for s in sl.segments], np.uint8)
        cases["det_pn64_%d" % ci] = np.array([[s.pixels_normalized[0].x, s.pixels_normalized[0].y, s.pixels_normalized[1].x,
                                               s.pixels_normalized[1].y] for s in sl.segments], np.float64).reshape(n, 4)
        cases["det_normal64_%d" % ci] = np.array([[s.normal.x, s.normal.y] for s in sl.segments], np.float64).reshape(n, 2)
        cases["det_wire%d" % ci] = np.frombuffer(ros_serialize(sl, "duckietown_msgs/SegmentList"), np.uint8)
        # what ground_projection_node receives is the message after the wire: float32 fields
        rx = SegmentList()
        rx.header = sl.header
        for s in sl.segments:
            t = Segment()
            t.color = s.color
            for e in range(2):
                t.pixels_normalized[e].x = float(np.float32(s.pixels_normalized[e].x))
                t.pixels_normalized[e].y = float(np.float32(s.pixels_normalized[e].y))
            t.normal.x, t.normal.y = float(np.float32(s.normal.x)), float(np.float32(s.normal.y))
            rx.segments.append(t)
        gnode.lineseglist_cb(rx)
        gl = gnode.pub_lineseglist_.msg
        assert len(gl.segments) == n
        cases["gp_color%d" % ci] = np.array([s.color for s in gl.segments], np.uint8)
        cases["gp_points%d" % ci] = np.array([[s.points[0].x, s.points[0].y, s.points[0].z, s.points[1].x, s.points[1].y, s.points[1].z]
                                              for s in gl.segments], np.float64).reshape(n, 6)
        cases["gp_wire%d" % ci] = np.frombuffer(ros_serialize(gl, "duckietown_msgs/SegmentList"), np.uint8)
    cases["n_cases"] = np.int32(len(geoms))
    np.savez_compressed(os.path.join(OUT, "node_pipeline.npz"), **cases)
    print("node_pipeline:", [int(cases["det_wire%d" % i].size) for i in range(len(geoms))], "wire bytes")


def golden_kmeans():
    """kmeans.py:22-47 on a few images: random pixels (exact distance ties in the first iteration, no convergence within
    max_iter), synthetic lane frames (one of the init colours missing: empty-cluster relocation), a noisy lane frame."""
    import contextlib, io, warnings
    warnings.filterwarnings("ignore")
    sys.path.insert(0, os.path.join(OUT, "..", ".."))
    from lane_slam_amd import synth
    km = load_file("ref_kmeans", os.path.join(REF, "anti_instagram/include/anti_instagram/kmeans.py"))
    rng = np.random.default_rng(5)
    imgs = {"random": np.random.default_rng(0).integers(0, 256, (120, 160, 3), dtype=np.uint8),
            "lane_a": synth.make_frame(3)[::2, ::2], "lane_b": synth.make_frame(11)[::4, ::4], "lane_c": synth.make_frame(20)[::4, ::4],
            "noisy": np.clip(synth.make_frame(5)[::2, ::2].astype(np.int32) + rng.integers(-40, 40, (240, 320, 3)), 0, 255).astype(np.uint8)}
    out = {"inits3": np.asarray(km.CENTERS, np.float64), "inits4": np.asarray(km.CENTERS2, np.float64)}
    for name, img in imgs.items():
        out["img_" + name] = img
        if name == "random":
            out["pts_" + name] = np.ascontiguousarray(km.getimgdatapts(img[-100:, :, :]))   # the reference's own point order (column major)
        for k, init in ((3, km.CENTERS), (4, km.CENTERS2)):
            with contextlib.redirect_stdout(io.StringIO()):
                centers, counts, score = km.runKMeans(img, k, init)
            out["centers%d_%s" % (k, name)] = np.asarray(centers, np.float64)
            out["counts%d_%s" % (k, name)] = np.array([int(counts[i]) for i in range(k)], np.int64)
            out["score%d_%s" % (k, name)] = np.float64(score)
    import sklearn
    out["sklearn_version"] = np.array(sklearn.__version__)
    np.savez_compressed(os.path.join(OUT, "kmeans.npz"), **out)
    print("kmeans.npz:", len(imgs), "images x 2 inits, scikit-learn", sklearn.__version__)


def golden_real_frames():
    """Three of the reference's real Duckiebot camera frames (src/anti_instagram/annotation-tool/images/*.jpg, data files
    of the reference, not source) as decoded B, G, R pixel arrays: INPUT data for the GPU parity tests, so that the
    front end is also compared with the oracle on camera images, not only on synthetic ones.  No expected outputs are
    stored: the oracle computes them at test time."""
    from PIL import Image
    d = os.path.join(REF, "anti_instagram/annotation-tool/images")
    names = sorted(os.listdir(d))
    picks = [names[0], names[len(names) // 2], names[-1]]
    out = {}
    for k, n in enumerate(picks):
        rgb = np.asarray(Image.open(os.path.join(d, n)).convert("RGB"))
        out["frame%d" % k] = np.ascontiguousarray(rgb[:, :, ::-1])       # B, G, R like cv2.imdecode
    np.savez_compressed(os.path.join(OUT, "real_frames.npz"), **out)
    print("real_frames.npz:", picks, [v.shape for v in out.values()])


def golden_real_jpegs(count=28):
    """`count` of the reference's 173 real Duckiebot camera frames (src/anti_instagram/annotation-tool/images/*.jpg: data files
    of the reference) as their JPEG BYTE STREAMS, spread evenly over the five recording sessions (night, noon, closed room, ice
    rink, coordination LEDs): input data for the GPU tests -- device JPEG decode -> the whole front end / the KeyLine path --
    and for bench.py's real-content rows.  No expected outputs are stored: the pinned JPEG oracle and the oracle front end
    compute them at test time."""
    d = os.path.join(REF, "anti_instagram/annotation-tool/images")
    names = sorted(n for n in os.listdir(d) if n.endswith(".jpg"))
    picks = [names[(2 * k + 1) * len(names) // (2 * count)] for k in range(count)]
    out = {"names": np.array(picks)}
    for k, n in enumerate(picks):
        with open(os.path.join(d, n), "rb") as f:
            out["jpeg%02d" % k] = np.frombuffer(f.read(), np.uint8)
    np.savez(os.path.join(OUT, "real_jpegs.npz"), **out)
    print("real_jpegs.npz:", count, "streams,", sum(v.size for k, v in out.items() if k != "names"), "bytes:", picks)


if __name__ == "__main__":
    install_stubs()
    golden_node_pipeline()
    install_stubs()
    golden_ground_projection()
    golden_find_normal()
    golden_line_sanity()
    golden_scaleandshift()
    install_stubs()
    golden_kmeans()
    golden_real_frames()
    golden_real_jpegs()
