"""ctypes binding of liblanefront.so (include/lanefront.h).  Fails loudly when the HIP
library has not been built: there is no Python or CPU fallback for any entry point."""
import ctypes
import os

from .config import LfConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
# LANEFRONT_LIBRARY points at an alternative build of the same HIP library (diagnostic builds such as
# -DLFG_STAMPS); the default is the in-tree product build.
SO_PATH = os.environ.get("LANEFRONT_LIBRARY") or os.path.join(_HERE, "liblanefront.so")

LF_N_STAGES = 14
LF_MAP_N_STAGES = 4
LF_MSG_DETECTOR, LF_MSG_GROUND, LF_MSG_FILTERED = 0, 1, 2
(LF_BUF_BGR, LF_BUF_MASKS, LF_BUF_EDGES, LF_BUF_LSD_ANGLE, LF_BUF_LSD_MODGRAD, LF_BUF_LSD_ORDER,
 LF_BUF_LSD_NORDER, LF_BUF_LBD_DX, LF_BUF_LBD_DY, LF_BUF_LSD_COUNTS, LF_BUF_LSD_SCRATCH, LF_BUF_LSD_NLOW) = range(12)

# every symbol include/lanefront.h declares
EXPORTS = (
    "lf_abi_version", "lf_create", "lf_destroy", "lf_last_error", "lf_synchronize", "lf_get_stream",
    "lf_set_image", "lf_detect_lines", "lf_process_batch", "lf_process_batch_async", "lf_wait", "lf_associate", "lf_associate_float", "lf_kmeans",
    "lf_jpeg_decode_batch", "lf_jpeg_info", "lf_frames_buffer", "lf_serialize_segments", "lf_deserialize_segments",
    "lf_debug_fetch", "lf_debug_detmath", "lf_debug_probe", "lf_debug_lsd_binary", "lf_lsd_size", "lf_set_profiling", "lf_get_timing", "lf_reset_timing", "lf_stage_name",
    "lf_map_create", "lf_map_destroy", "lf_map_last_error", "lf_map_get_stream", "lf_map_synchronize", "lf_map_seed", "lf_map_size",
    "lf_map_associate", "lf_map_pack_block", "lf_map_update", "lf_map_step", "lf_map_step_host", "lf_map_fetch",
    "lf_map_set_profiling", "lf_map_get_timing", "lf_map_stage_name",
    "lf_descriptor_default_params", "lf_set_descriptor_params", "lf_get_descriptor_params",
    "lf_edlines_default_params", "lf_keylines_batch", "lf_describe_keylines", "lf_keylines_debug_fetch", "lf_set_image_edlines", "lf_knn_match", "lf_radius_match", "lf_jpeg_decode_batch_gpu", "lf_jpeg_decode_batch_gpu_async", "lf_jpeg_status", "lf_jpeg_decode_for_detect_async",
    "lf_set_tie_rule", "lf_map_set_tie_rule", "lf_debug_std_sort", "lf_suggested_depth", "lf_lsd_list_capacity", "lf_lsd_scratch_stride", "lf_set_detector", "lf_detector_failures", "lf_keylines_batch_async", "lf_keylines_frame_status", "lf_lsd_keylines_batch", "lf_select_queries",
    "lf_lsd_default_options", "lf_lsd_keylines_batch_ex", "lf_keylines_batch_masked",
    "lf_matcher_add", "lf_matcher_clear", "lf_matcher_size", "lf_matcher_match", "lf_matcher_knn_match", "lf_matcher_radius_match",
)
DETECTORS = {"lsd": 0, "edlines": 1}
TIE_RULES = {"lowest": 0, "mihasher": 1}
LF_MAX_OCTAVES = 5


class LfSegments(ctypes.Structure):
    _fields_ = [
        ("capacity", ctypes.c_int32),
        ("frame_offset", ctypes.c_void_p),
        ("lines", ctypes.c_void_p), ("normals", ctypes.c_void_p), ("color", ctypes.c_void_p),
        ("pixels_normalized", ctypes.c_void_p), ("ground", ctypes.c_void_p), ("keep", ctypes.c_void_p),
        ("desc", ctypes.c_void_p), ("code", ctypes.c_void_p),
    ]


class LfEdlinesParams(ctypes.Structure):
    """ctypes mirror of `lf_edlines_params` (include/lanefront.h)."""
    _fields_ = [("gradient_threshold", ctypes.c_int32), ("anchor_threshold", ctypes.c_int32), ("scan_intervals", ctypes.c_int32),
                ("min_line_len", ctypes.c_int32), ("line_fit_err_threshold", ctypes.c_double), ("ksize", ctypes.c_int32)]


class LfDescriptorParams(ctypes.Structure):
    """lf_descriptor_params: BinaryDescriptor::Params (binary_descriptor_custom.cpp:108-116)."""
    _fields_ = [("num_of_octave", ctypes.c_int32), ("width_of_band", ctypes.c_int32), ("reduction_ratio", ctypes.c_int32), ("ksize", ctypes.c_int32)]


KEYLINE_FIELDS = (("start_end", "f4", 4), ("in_octave", "f4", 4), ("angle", "f4", 1), ("num_pixels", "i4", 1), ("line_length", "f4", 1),
                  ("octave", "i4", 1), ("class_id", "i4", 1), ("response", "f4", 1), ("size", "f4", 1), ("pt", "f4", 2), ("salience", "f4", 1),
                  ("desc", "f4", 72), ("code", "u1", 32))


class LfKeylines(ctypes.Structure):
    """ctypes mirror of `lf_keylines` (include/lanefront.h)."""
    _fields_ = [("capacity", ctypes.c_int32), ("frame_offset", ctypes.c_void_p)] + [(k, ctypes.c_void_p) for k, _, _ in KEYLINE_FIELDS]


class LfLsdOptions(ctypes.Structure):
    """ctypes mirror of `lf_lsd_options` (include/lanefront.h): LSDDetectorC::LSDOptions (descriptor_custom.hpp:906-916)."""
    _fields_ = [("refine", ctypes.c_int32), ("n_bins", ctypes.c_int32), ("scale", ctypes.c_double), ("sigma_scale", ctypes.c_double),
                ("quant", ctypes.c_double), ("ang_th", ctypes.c_double), ("log_eps", ctypes.c_double), ("density_th", ctypes.c_double),
                ("min_length", ctypes.c_double)]


class LfMapConfig(ctypes.Structure):
    """ctypes mirror of `lf_map_config` (include/lanefront.h)."""
    _fields_ = [(k, ctypes.c_int32) for k in ("capacity", "color_gating", "max_distance", "policy", "kept_only",
                                              "merge_distance", "when_full")]


_lib = None


def load():
    """Load liblanefront.so (built by `make -C lane_slam_amd/csrc` or __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(
            "lanefront: %s is missing. Build the HIP library first (python -c 'import __graft_entry__ as g; "
            "g.build()' or make -C lane_slam_amd/csrc). There is no CPU fallback." % SO_PATH)
    # kernel arguments in device memory (about 2 us less per launch: INTEGRATION.md section 4); only a default, and only
    # effective when the HIP runtime has not been initialised by someone else before
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    lib = ctypes.CDLL(SO_PATH)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    lib.lf_abi_version.restype = ci
    lib.lf_create.argtypes = [ctypes.POINTER(LfConfig), ci, ci, ci, ctypes.POINTER(vp)]
    lib.lf_create.restype = ci
    lib.lf_destroy.argtypes = [vp]
    lib.lf_destroy.restype = None
    lib.lf_last_error.argtypes = [vp]
    lib.lf_last_error.restype = ctypes.c_char_p
    lib.lf_synchronize.argtypes = [vp]
    lib.lf_get_stream.argtypes = [vp, ctypes.POINTER(vp)]
    lib.lf_get_stream.restype = ci
    lib.lf_set_image.argtypes = [vp, vp, ci, ci, ci]
    lib.lf_detect_lines.argtypes = [vp, ci, vp, vp, vp, vp, ci, ctypes.POINTER(ci)]
    lib.lf_process_batch.argtypes = [vp, vp, ci, ci, ctypes.POINTER(LfSegments), ci, ci, ctypes.POINTER(ci)]
    lib.lf_process_batch_async.argtypes = [vp, vp, ci, ci, ctypes.POINTER(LfSegments), ci]
    lib.lf_process_batch_async.restype = ci
    lib.lf_wait.argtypes = [vp, ctypes.POINTER(ci)]
    lib.lf_wait.restype = ci
    lib.lf_associate.argtypes = [vp, vp, ci, vp, ci, vp, vp, ci]
    lib.lf_associate_float.argtypes = [vp, vp, ci, vp, ci, vp, vp, ci]
    lib.lf_kmeans.argtypes = [vp, vp, ci, ci, ci, vp, ci, ctypes.c_double, vp, vp, vp, vp]
    lib.lf_jpeg_decode_batch.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t), ci, ci, ci, vp, ci, ci,
                                         ctypes.POINTER(ci)]
    lib.lf_jpeg_decode_batch.restype = ci
    lib.lf_jpeg_decode_batch_gpu.argtypes = lib.lf_jpeg_decode_batch.argtypes
    lib.lf_jpeg_decode_batch_gpu.restype = ci
    lib.lf_jpeg_decode_batch_gpu_async.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t), ci, ci, ci, vp, ci]
    lib.lf_jpeg_decode_batch_gpu_async.restype = ci
    lib.lf_jpeg_status.argtypes = [vp, ctypes.POINTER(ci), ci, ctypes.POINTER(ci)]
    lib.lf_jpeg_decode_for_detect_async.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t), ci, ci]
    lib.lf_jpeg_decode_for_detect_async.restype = ci
    lib.lf_jpeg_status.restype = ci
    lib.lf_jpeg_info.argtypes = [vp, ctypes.c_size_t] + [ctypes.POINTER(ci)] * 5
    lib.lf_jpeg_info.restype = ci
    lib.lf_frames_buffer.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t)]
    lib.lf_frames_buffer.restype = ci
    lib.lf_serialize_segments.argtypes = [vp, ctypes.POINTER(LfSegments), ci, ci, ci, vp, ctypes.c_size_t, ci,
                                          ctypes.POINTER(ctypes.c_int64)]
    lib.lf_serialize_segments.restype = ci
    lib.lf_deserialize_segments.argtypes = [vp, vp, ci, ctypes.POINTER(ctypes.c_int64), ci, ctypes.POINTER(LfSegments), ci,
                                            ctypes.POINTER(ci)]
    lib.lf_deserialize_segments.restype = ci
    lib.lf_debug_fetch.argtypes = [vp, ci, vp, ctypes.c_size_t]
    lib.lf_debug_detmath.argtypes = [vp, ci, vp, vp, vp, ci]
    lib.lf_debug_detmath.restype = ci
    lib.lf_debug_probe.argtypes = [vp, ci, ci, ctypes.c_size_t, ci]
    lib.lf_debug_probe.restype = ci
    lib.lf_debug_lsd_binary.argtypes = [vp, vp, ci, ci, vp, ci, ctypes.POINTER(ci)]
    lib.lf_debug_lsd_binary.restype = ci
    lib.lf_lsd_size.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(ci)]
    lib.lf_set_profiling.argtypes = [vp, ci]
    lib.lf_get_timing.argtypes = [vp, vp, vp, ci]
    lib.lf_reset_timing.argtypes = [vp]
    lib.lf_stage_name.argtypes = [ci]
    lib.lf_stage_name.restype = ctypes.c_char_p
    i64p = ctypes.POINTER(ctypes.c_int64)
    lib.lf_map_create.argtypes = [ci, ctypes.POINTER(LfMapConfig), ctypes.POINTER(vp)]
    lib.lf_map_destroy.argtypes = [vp]
    lib.lf_map_destroy.restype = None
    lib.lf_map_last_error.argtypes = [vp]
    lib.lf_map_last_error.restype = ctypes.c_char_p
    lib.lf_map_get_stream.argtypes = [vp, ctypes.POINTER(vp)]
    lib.lf_map_synchronize.argtypes = [vp]
    lib.lf_map_seed.argtypes = [vp, vp, vp, vp, ci, ci]
    lib.lf_map_size.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(ci), i64p, i64p]
    lib.lf_map_associate.argtypes = [vp, vp, vp, vp, ci, vp, vp, ci]
    lib.lf_map_pack_block.argtypes = [vp, vp, ctypes.POINTER(LfSegments), ci, ci, vp, vp, vp, ci, vp, ci]
    lib.lf_map_update.argtypes = [vp, vp, ci, ci]
    lib.lf_map_step.argtypes = [vp, vp, ctypes.POINTER(LfSegments), ci, ci, vp, ci, vp, vp]
    lib.lf_map_step_host.argtypes = [vp, ctypes.POINTER(LfSegments), ci, ci, vp, ci, vp, vp]
    lib.lf_map_step_host.restype = ci
    lib.lf_map_fetch.argtypes = [vp, ci, ci, vp, vp, vp, vp, vp]
    lib.lf_map_set_profiling.argtypes = [vp, ci]
    lib.lf_map_set_profiling.restype = ci
    lib.lf_map_get_timing.argtypes = [vp, vp, vp, ci]
    lib.lf_map_get_timing.restype = ci
    lib.lf_map_stage_name.argtypes = [ci]
    lib.lf_map_stage_name.restype = ctypes.c_char_p
    lib.lf_descriptor_default_params.argtypes = [ctypes.POINTER(LfDescriptorParams)]
    lib.lf_descriptor_default_params.restype = None
    lib.lf_set_descriptor_params.argtypes = [vp, ctypes.POINTER(LfDescriptorParams)]
    lib.lf_set_descriptor_params.restype = ci
    lib.lf_get_descriptor_params.argtypes = [vp, ctypes.POINTER(LfDescriptorParams)]
    lib.lf_get_descriptor_params.restype = ci
    lib.lf_edlines_default_params.argtypes = [ctypes.POINTER(LfEdlinesParams)]
    lib.lf_edlines_default_params.restype = None
    lib.lf_keylines_batch.argtypes = [vp, vp, ci, ci, ci, ci, ctypes.POINTER(LfEdlinesParams), ctypes.POINTER(LfKeylines), ci, ci,
                                      ctypes.POINTER(ci), vp]
    lib.lf_keylines_batch.restype = ci
    lib.lf_keylines_batch_masked.argtypes = [vp, vp, ci, ci, ci, ci, ctypes.POINTER(LfEdlinesParams), vp, ci, ctypes.POINTER(LfKeylines), ci, ci,
                                             ctypes.POINTER(ci), vp]
    lib.lf_keylines_batch_masked.restype = ci
    lib.lf_describe_keylines.argtypes = [vp, vp, ci, vp, vp, vp, vp, vp, ci, vp, vp, ci]
    lib.lf_describe_keylines.restype = ci
    lib.lf_keylines_debug_fetch.argtypes = [vp, ci, ci, vp, ctypes.c_size_t, vp]
    lib.lf_keylines_debug_fetch.restype = ci
    lib.lf_set_image_edlines.argtypes = [vp, vp, ci, ci, ci, ctypes.POINTER(LfEdlinesParams)]
    lib.lf_set_image_edlines.restype = ci
    lib.lf_keylines_batch_async.argtypes = [vp, vp, ci, ci, ci, ctypes.POINTER(LfEdlinesParams), ctypes.POINTER(LfKeylines), ci]
    lib.lf_keylines_batch_async.restype = ci
    lib.lf_select_queries.argtypes = [vp, vp, ci, vp, vp, vp, ctypes.POINTER(ci), ci]
    lib.lf_select_queries.restype = ci
    lib.lf_matcher_add.argtypes = [vp, vp, ci, ci]
    lib.lf_matcher_clear.argtypes = [vp]
    lib.lf_matcher_size.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(ci)]
    lib.lf_matcher_match.argtypes = [vp, vp, ci, vp, vp, ctypes.POINTER(ci)]
    lib.lf_matcher_knn_match.argtypes = [vp, vp, ci, ci, vp, ci, vp, vp, ctypes.POINTER(ci)]
    lib.lf_matcher_radius_match.argtypes = [vp, vp, ci, ctypes.c_float, vp, ci, vp, vp, ci, ctypes.POINTER(ci), ctypes.POINTER(ci)]
    for fn in (lib.lf_matcher_add, lib.lf_matcher_clear, lib.lf_matcher_size, lib.lf_matcher_match, lib.lf_matcher_knn_match, lib.lf_matcher_radius_match):
        fn.restype = ci
    lib.lf_lsd_keylines_batch.argtypes = [vp, vp, ci, ci, ci, ci, ctypes.POINTER(LfKeylines), ci, ci, ctypes.POINTER(ci)]
    lib.lf_lsd_keylines_batch.restype = ci
    lib.lf_lsd_keylines_batch_ex.argtypes = [vp, vp, ci, ci, ci, ci, ctypes.POINTER(LfLsdOptions), vp, ci, ctypes.POINTER(LfKeylines), ci, ci, ctypes.POINTER(ci)]
    lib.lf_lsd_keylines_batch_ex.restype = ci
    lib.lf_lsd_default_options.argtypes = [ctypes.POINTER(LfLsdOptions)]
    lib.lf_lsd_default_options.restype = None
    lib.lf_keylines_frame_status.argtypes = [vp, vp, ci]
    lib.lf_keylines_frame_status.restype = ci
    lib.lf_set_detector.argtypes = [vp, ci, ctypes.POINTER(LfEdlinesParams)]
    lib.lf_set_detector.restype = ci
    lib.lf_detector_failures.argtypes = [vp]
    lib.lf_detector_failures.restype = ci
    lib.lf_suggested_depth.argtypes = [vp]
    lib.lf_lsd_list_capacity.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(ci)]
    lib.lf_lsd_list_capacity.restype = ci
    lib.lf_lsd_scratch_stride.argtypes = [vp]
    lib.lf_lsd_scratch_stride.restype = ci
    lib.lf_suggested_depth.restype = ci
    lib.lf_debug_std_sort.argtypes = [vp, vp, ci, vp]
    lib.lf_debug_std_sort.restype = ci
    lib.lf_set_tie_rule.argtypes = [vp, ci]
    lib.lf_set_tie_rule.restype = ci
    lib.lf_map_set_tie_rule.argtypes = [vp, ci]
    lib.lf_map_set_tie_rule.restype = ci
    lib.lf_knn_match.argtypes = [vp, vp, ci, vp, ci, ci, vp, vp, ci]
    lib.lf_knn_match.restype = ci
    lib.lf_radius_match.argtypes = [vp, vp, ci, vp, ci, ctypes.c_float, vp, vp, vp, ci, ctypes.POINTER(ci), ci]
    lib.lf_radius_match.restype = ci
    for f in ("lf_map_create", "lf_map_get_stream", "lf_map_synchronize", "lf_map_seed", "lf_map_size", "lf_map_associate",
              "lf_map_pack_block", "lf_map_update", "lf_map_step", "lf_map_fetch"):
        getattr(lib, f).restype = ci
    for f in ("lf_synchronize", "lf_set_image", "lf_detect_lines", "lf_process_batch", "lf_process_batch_async", "lf_wait", "lf_associate",
              "lf_associate_float", "lf_kmeans", "lf_jpeg_decode_batch", "lf_jpeg_info", "lf_frames_buffer", "lf_serialize_segments", "lf_deserialize_segments",
    "lf_debug_fetch", "lf_debug_detmath", "lf_debug_lsd_binary", "lf_lsd_size", "lf_set_profiling", "lf_get_timing",
              "lf_reset_timing"):
        getattr(lib, f).restype = ci
    _lib = lib
    return lib
