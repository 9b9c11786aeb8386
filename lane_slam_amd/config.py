"""Configuration of the line-feature front end.

Every default below is a constant of the reference, cited by file:line
(paths relative to /root/reference):

* geometry            src/duckietown/config/baseline/line_detector/line_detector_node/default.yaml:1-2
* detector thresholds same file :11-23
* LSD parameters      cv2.createLineSegmentDetector defaults behind
                      src/line_detector/include/line_detector/line_detector_lsd.py:65
* AntiInstagram       src/anti_instagram/include/anti_instagram/AntiInstagram.py:86-89 (identity)
* homography          src/duckietown/include/calibrations/camera_extrinsic/default.yaml:1
* intrinsics          src/duckietown/include/calibrations/camera_intrinsic/default.yaml:1-22
* sanity constants    src/line_sanity/src/line_sanity_node.py:17-23
"""
import copy
import ctypes

# LSD seed order inside a gradient bin (include/lanefront.h, lf_config.lsd_seed_order): OpenCV 3.0 / 3.1 keep raster order,
# 3.2 ... 3.4.5 (ROS Kinetic's 3.3.1) leave what std::sort leaves
LSD_SEED_ORDERS = {"opencv30": 0, "opencv32": 1}
WHITE, YELLOW, RED = 0, 1, 2          # src/duckietown_msgs/msg/Segment.msg:1-3
COLOR_NAMES = ("white", "yellow", "red")

# the 13 keys LineDetectorLSD accepts (line_detector_lsd.py:20-34)
DETECTOR_KEYS = (
    "hsv_white1", "hsv_white2", "hsv_yellow1", "hsv_yellow2",
    "hsv_red1", "hsv_red2", "hsv_red3", "hsv_red4",
    "dilation_kernel_size", "canny_thresholds",
    "hough_threshold", "hough_min_line_length", "hough_max_line_gap",
)

DEFAULT_DETECTOR_CONFIGURATION = {
    "dilation_kernel_size": 3,
    "canny_thresholds": [80, 200],
    "hough_threshold": 2,
    "hough_min_line_length": 3,
    "hough_max_line_gap": 1,
    "hsv_white1": [0, 0, 150],
    "hsv_white2": [180, 60, 255],
    "hsv_yellow1": [25, 140, 100],
    "hsv_yellow2": [45, 255, 255],
    "hsv_red1": [0, 140, 100],
    "hsv_red2": [15, 255, 255],
    "hsv_red3": [165, 140, 100],
    "hsv_red4": [180, 255, 255],
}

DEFAULT_HOMOGRAPHY = [-4.89775e-05, -0.0002150858, -0.1818273,
                      0.00099274, 1.202336e-06, -0.3280241,
                      -0.0004281805, -0.007185673, 1.0]
DEFAULT_K = [307.7379294605756, 0, 329.692367951685,
             0, 314.9827773443905, 244.4605588877848,
             0, 0, 1]
DEFAULT_D = [-0.2565888993516047, 0.04481160508242147, -0.00505275149956019,
             0.001308569367976665, 0]
DEFAULT_R = [1, 0, 0, 0, 1, 0, 0, 0, 1]
DEFAULT_P = [210.1107940673828, 0, 327.2577820024981, 0,
             0, 253.8408660888672, 239.9969353923052, 0,
             0, 0, 1, 0]


def default_config(geometry="parity", in_size=(480, 640)):
    """Plain-dict configuration.

    geometry: "parity"  -> img_size [120,160], top_cutoff 40 (the reference default)
              "fullres" -> img_size = in_size, top_cutoff = in_rows/3 (640x320 working image)
    """
    in_rows, in_cols = in_size
    if geometry == "parity":
        img_size, top_cutoff = [120, 160], 40
    elif geometry == "fullres":
        img_size, top_cutoff = [in_rows, in_cols], in_rows // 3
    else:
        raise ValueError("unknown geometry %r" % (geometry,))
    cfg = {
        "in_size": [in_rows, in_cols],
        "img_size": img_size,
        "top_cutoff": top_cutoff,
        "ai_scale": [1.0, 1.0, 1.0],
        "ai_shift": [0.0, 0.0, 0.0],
        "detector": copy.deepcopy(DEFAULT_DETECTOR_CONFIGURATION),
        "lsd": {"refine": 2, "scale": 0.8, "sigma_scale": 0.6, "quant": 2.0, "ang_th": 22.5,
                "log_eps": 0.0, "density_th": 0.7, "n_bins": 1024,
                # the seed order inside a gradient bin: "opencv32" = std::sort's (OpenCV 3.2 .. 3.4.5: the reference's ROS Kinetic
                # ships 3.3.1), "opencv30" = raster order (3.0 / 3.1; about 0.3 ms less per 256-frame batch)
                "seed_order": "opencv32"},
        "H": list(DEFAULT_HOMOGRAPHY), "K": list(DEFAULT_K), "D": list(DEFAULT_D),
        "R": list(DEFAULT_R), "P": list(DEFAULT_P),
        "cam_size": [480, 640],
        "sanity": {"lanewidth": 0.23, "linewidth_white": 0.05, "linewidth_yellow": 0.025,
                   "d_min": -0.15, "d_max": 0.3, "phi_min": -1.5, "phi_max": 1.5},
    }
    return cfg


class LfConfig(ctypes.Structure):
    """ctypes mirror of `lf_config` (include/lanefront.h)."""
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


def fill_struct(s, cfg):
    """Copy a config dict into a ctypes struct with the lf_config field layout."""
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
    order = lsd.get("seed_order", "opencv32")
    if order not in LSD_SEED_ORDERS:
        raise ValueError("lsd.seed_order must be one of %r" % (sorted(LSD_SEED_ORDERS),))
    s.lsd_seed_order = LSD_SEED_ORDERS[order]
    return s


def work_size(cfg):
    """(rows, cols) of the working image after resize + crop (line_detector_node.py:163-169)."""
    return cfg["img_size"][0] - cfg["top_cutoff"], cfg["img_size"][1]
