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
        self.points = [_Obj(x=0.0, y=0.0, z=0.0), _Obj(x=0.0, y=0.0, z=0.0)]


class SegmentList(object):
    def __init__(self):
        self.header = None
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


if __name__ == "__main__":
    install_stubs()
    golden_ground_projection()
    golden_find_normal()
    golden_line_sanity()
    golden_scaleandshift()
