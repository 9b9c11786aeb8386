/*
 * lanefront -- MI355X-native line-feature front end for lane-slam.  C ABI.
 *
 * The reference has no FFI: its hot path is Python calling cv2 / numpy, plus an
 * unbuilt C++ library.  Each entry point below names the reference interface it
 * replaces (paths relative to /root/reference).  INTEGRATION.md shows the
 * ctypes binding a maintainer adds on the reference side.
 *
 * Conventions: every function returns an lf_status (0 = ok, < 0 = error) and
 * never throws or aborts across the ABI; lf_last_error() returns a static,
 * NUL-terminated description of the last failure on that handle.  Pointers are
 * plain host or device addresses as stated per argument; nothing returned by
 * the library is owned by it.  A handle is NOT thread-safe: one in-flight
 * frame/batch per handle, exactly like the reference node, which holds a
 * non-blocking lock around its detector
 * (src/line_detector/src/line_detector_node.py:129-139).
 * There is no CPU fallback: without a HIP device lf_create fails with
 * LF_ERR_HIP.
 *
 * OpenCV version.  The reference calls cv2 for resize, convertScaleAbs, BGR2HSV, inRange, dilate, Canny, the line
 * segment detector, GaussianBlur, Sobel, BGR2GRAY and (through image_geometry) undistortPoints without pinning a
 * version (src/line_descriptor/CMakeLists.txt:17 `find_package(OpenCV 3 REQUIRED)`).  The arithmetic restated here is
 * that of OpenCV 3.0 - 3.3 as ROS Kinetic / Melodic ship it (the f64 LSD with REFINE_ADV; 8-bit fixed-point HSV; the
 * 5x5 sigma-1 8-bit Gaussian taps {14, 63, 103, 63, 14} / 256).  OpenCV >= 3.4.7 / 4.1.1 changed the 8-bit Gaussian
 * taps, OpenCV >= 4.0 builds without the LSD; results against such a cv2 differ.  No real OpenCV exists in the build
 * image, so these stages are "parity unpinned" (DESIGN.md section 2).
 */
#ifndef LANEFRONT_H
#define LANEFRONT_H

#include <stddef.h>
#include <stdint.h>

/* the library is built with -fvisibility=hidden: exactly the entry points declared here are exported */
#define LF_API __attribute__((visibility("default")))

#ifdef __cplusplus
extern "C" {
#endif

#define LF_ABI_VERSION 5   /* 2: JPEG ingest, SegmentList glue, LF_ERR_DECODE, 13 timing stages; 3: live map (lf_map_*); 4: EDLines / KeyLines, block overflow marker; 5: lf_config.lsd_seed_order, tie rules */

typedef enum lf_status {
    LF_OK = 0,
    LF_ERR_BAD_ARG = -1,
    LF_ERR_CAPACITY = -2,     /* caller buffer / configured capacity exceeded */
    LF_ERR_HIP = -3,          /* HIP runtime error or no device */
    LF_ERR_NOT_INITIALISED = -4,
    LF_ERR_UNSUPPORTED = -5,
    LF_ERR_DECODE = -6        /* undecodable JPEG stream (the reference gets None from cv2.imdecode and drops the frame) */
} lf_status;

/* colour codes: src/duckietown_msgs/msg/Segment.msg:1-3 */
#define LF_WHITE 0
#define LF_YELLOW 1
#define LF_RED 2

/*
 * Configuration.  Field sources:
 *   in_*, img_*, top_cutoff  line_detector_node.py:163-169, default.yaml:1-2
 *   ai_scale / ai_shift      anti_instagram/scale_and_shift.py:25-33 (BGR channel order)
 *   hsv_*, dilation, canny   line_detector_lsd.py:20-34,38-62; default.yaml:11-23
 *                            box 0 white, 1 yellow, 2 red1..red2, 3 red3..red4
 *   lsd_*                    cv2.createLineSegmentDetector arguments (line_detector_lsd.py:65)
 *   H, K, D, R, P, cam_*     ground_projection/GroundProjection.py:38-78,143-157
 *   sanity constants         line_sanity/src/line_sanity_node.py:17-23
 */
typedef struct lf_config {
    int32_t in_rows, in_cols;
    int32_t img_rows, img_cols;
    int32_t top_cutoff;
    float ai_scale[3], ai_shift[3];
    int32_t hsv_lo[4][3], hsv_hi[4][3];
    int32_t dilation_kernel_size;
    double canny_lo, canny_hi;
    int32_t lsd_refine;            /* 0 none, 1 standard, 2 advanced (reference uses 2) */
    int32_t lsd_n_bins;
    double lsd_scale, lsd_sigma_scale, lsd_quant, lsd_ang_th, lsd_log_eps, lsd_density_th;
    double H[9], K[9], D[5], R[9], P[12];
    int32_t cam_w, cam_h;
    double lanewidth, linewidth_white, linewidth_yellow, d_min, d_max, phi_min, phi_max;
    /* Which OpenCV's LSD seed order (line_detector_lsd.py:64-72 calls cv2's detector; region growing depends on the order in
     * which equally strong pixels seed regions):
     *   LF_LSD_SEED_OPENCV30  3.0 / 3.1: per-bin lists, raster order inside a gradient bin (one counting sort).
     *   LF_LSD_SEED_OPENCV32  3.2 ... 3.4.5 -- ROS Kinetic's 3.3.1, the stack the reference names (README.md:54): every pixel of
     *                         the gradient image sorted with std::sort(compare_norm); inside a bin the order is what libstdc++'s
     *                         introsort leaves, reproduced on the device move for move (k_lsd_seed32.hip; any working image up
     *                         to 2^21 LSD pixels since round 5).  THIS IS WHAT THE REFERENCE'S STACK COMPUTES: a caller that wants the
     *                         reference's segments sets it (the Python mirror's default_config() and plugin classes do); the
     *                         value 0 of a zeroed struct is the older order, kept as an A/B option. */
    int32_t lsd_seed_order;
    int32_t reserved0;             /* 0 */
} lf_config;
#define LF_LSD_SEED_OPENCV30 0
#define LF_LSD_SEED_OPENCV32 1

typedef struct lf_handle lf_handle;

/* Struct-of-arrays segment block (the SegmentList of a batch).  Every array is
 * caller-allocated with room for `capacity` segments; NULL arrays are skipped.
 * Segments are ordered by frame, then white, yellow, red, then detection order
 * (line_detector_node.py:197-205).  Field meaning per segment:
 *   lines              x1,y1,x2,y2 in working-image pixels after endpoint ordering (line_detector_lsd.py:79-84)
 *   normals            Segment.normal (float32)                     (line_detector_node.py:262-263)
 *   color              Segment.color
 *   pixels_normalized  Segment.pixels_normalized[0..1]              (line_detector_node.py:195-205)
 *   ground             Segment.points[0..1].(x,y); z is 0           (ground_projection_node.py:60-61)
 *   keep               1 if LineSanityNode.processSegmentList keeps it (line_sanity_node.py:48-72)
 *   desc / code        float (72) / binary (32 B) LBD descriptor    (binary_descriptor_custom.cpp:1026-1372,653-667)
 *   frame_offset       n_frames+1 prefix offsets into the arrays
 */
typedef struct lf_segments {
    int32_t capacity;
    int32_t* frame_offset;
    float* lines;
    float* normals;
    uint8_t* color;
    float* pixels_normalized;
    double* ground;
    uint8_t* keep;
    float* desc;
    uint8_t* code;
} lf_segments;

/* ---- lifetime ------------------------------------------------------------ */
LF_API int lf_abi_version(void);

/* max_frames: largest batch lf_process_batch will be given (1 for the plugin path).
 * max_lines_per_color: capacity of one LSD run (one frame, one colour). */
LF_API int lf_create(const lf_config* cfg, int device_id, int max_frames, int max_lines_per_color,
              lf_handle** out);
LF_API void lf_destroy(lf_handle* h);
LF_API const char* lf_last_error(const lf_handle* h);
/* wait for all work queued on the handle's HIP stream */
LF_API int lf_synchronize(lf_handle* h);
/* the handle's HIP stream (a hipStream_t) for callers that order their own device work against the handle's with
 * events instead of host synchronisation (e.g. torch.cuda.ExternalStream); the stream stays owned by the handle */
LF_API int lf_get_stream(lf_handle* h, void** hip_stream);

/* ---- plugin path: replaces LineDetectorLSD (line_detector_lsd.py:11-142) ---
 * lf_set_image  <-> LineDetectorLSD.setImage(bgr)      (:135-139)
 * lf_detect_lines <-> LineDetectorLSD.detectLines(color) (:127-133): returns
 *   lines (n,4) float32 reordered, normals (n,2) float64, centers (n,2) float32,
 *   area = dilated colour mask (rows*cols u8, may be NULL).
 * The image is the already resized / cropped / colour-corrected working image
 * (line_detector_node.py:180); it must be rows x cols == the handle's working
 * size (img_rows - top_cutoff, img_cols).  Host pointers.
 */
LF_API int lf_set_image(lf_handle* h, const uint8_t* bgr, int rows, int cols, int row_stride_bytes);
LF_API int lf_detect_lines(lf_handle* h, int color, float* lines4, double* normals2, float* centers2,
                    uint8_t* area_or_null, int cap, int* n_out);

/* ---- batch path: processImage_ + ground_projection + line_sanity + describe -
 * Replaces, for a batch of raw camera frames (n_frames x in_rows x in_cols x 3, BGR u8):
 *   line_detector_node.py:163-213, ground_projection_node.py:55-65,
 *   line_sanity_node.py:48-72, and BinaryDescriptor::compute
 *   (binary_descriptor_custom.cpp:524-687) on the detected segments.
 * frames_on_device / out_on_device: 0 = host pointers, 1 = device pointers
 * (all arrays of `out` alike).  n_segments receives the total (host int).
 * Host frames are copied to the device one batch per call, and only the source
 * rows the working image reads (the rows from top_cutoff down; the crop of
 * line_detector_node.py:169 happens before anything else looks at a pixel).
 * With device outputs the call is asynchronous except for the final count
 * read-back; with host outputs it returns when the data is in place.
 */
LF_API int lf_process_batch(lf_handle* h, const uint8_t* frames, int n_frames, int frames_on_device,
                     lf_segments* out, int out_on_device, int describe, int* n_segments);

/* Pipelined form: lf_process_batch_async queues the whole batch on the handle's HIP stream and
 * returns at once (device outputs only); lf_wait blocks until it is done and returns the
 * segment count.  Several handles used in turn keep as many independent batches in flight, which
 * lets one batch's latency-bound LSD region growing overlap the next batch's streaming
 * kernels.  One batch in flight per handle.  Device frames and the out_dev arrays must stay valid and unchanged until lf_wait
 * returns: when a batch needs longer per-problem lists than the handle holds, lf_wait grows them and runs the batch a second
 * time from the same inputs into the same outputs (lf_lsd_list_capacity). */
LF_API int lf_process_batch_async(lf_handle* h, const uint8_t* frames, int n_frames, int frames_on_device,
                           lf_segments* out_dev, int describe);
LF_API int lf_wait(lf_handle* h, int* n_segments);

/* Which detector lf_process_batch / lf_process_batch_async run for stages a-2 .. a-4 of this handle's batches:
 *   LF_DETECTOR_LSD      (default) the reference's: 3-channel Canny, colour masks, cv2 LSD (line_detector_lsd.py:38-72)
 *   LF_DETECTOR_EDLINES  the package's second LineDetectorInterface implementation (lf_set_image_edlines is
 *                        its one-frame form, same contract): EDLines (BinaryDescriptor::detect's detector,
 *                        binary_descriptor_custom.cpp:415-513, one octave) on BGR2GRAY of the working image; a line belongs
 *                        to every colour whose dilated mask is set under its truncated, clamped centre; then the SAME
 *                        _findNormal / ordering, projection, line sanity and LBD stages as the LSD path, pipelined the same
 *                        way (no host synchronisation before lf_wait).  A frame on which EDLines gives up (its anchor / edge /
 *                        line arrays full: the reference prints "Line Detection not finished" and returns no lines) has no
 *                        segments; lf_detector_failures tells how many frames of the last completed batch did.
 * params: NULL = lf_edlines_default_params.  Not while a batch is in flight. */
#define LF_DETECTOR_LSD 0
#define LF_DETECTOR_EDLINES 1
struct lf_edlines_params;
LF_API int lf_set_detector(lf_handle* h, int detector, const struct lf_edlines_params* params_or_null);
LF_API int lf_detector_failures(const lf_handle* h);

/* ---- association: replaces BinaryDescriptorMatcher::match ------------------
 * (binary_descriptor_matcher.cpp:197-254): exact Hamming nearest neighbour of
 * each 256-bit query code in the map; idx = -1 and dist = -1 when the nearest
 * neighbour is farther than 128 bits (:721).
 * Computed as an exact matrix-core contraction (FP4 e2m1 +-1 operands with f32 accumulation on gfx950), one kernel launch per call.  on_device applies to all four arrays.
 *
 * Which of several EQUALLY near map codes is returned (the distance never depends on it) is the tie rule:
 *   LF_TIE_MIHASHER  the reference's: the candidate Mihasher::query meets first (binary_descriptor_matcher.cpp:635-753) --
 *                    by search radius, then substring, then the position of the differing bits in its enumeration
 *                    (:681-741), then train index.  Costs a second matrix pass that ranks only the ties (k_assoc_ties.hip).
 *                    THE DEFAULT of lf_associate and of the live map (lf_map_*): it is what BinaryDescriptorMatcher::match returns.
 *   LF_TIE_LOWEST    the lowest map index; one pass (an A/B option; the default of the live map until round 4).
 */
#define LF_TIE_LOWEST 0
#define LF_TIE_MIHASHER 1
LF_API int lf_associate(lf_handle* h, const uint8_t* query32, int nq, const uint8_t* map32, int nm,
                 int32_t* idx, float* dist, int on_device);
/* tie rule of this handle's lf_associate, lf_knn_match and lf_radius_match (LF_TIE_MIHASHER unless set) */
LF_API int lf_set_tie_rule(lf_handle* h, int tie_rule);
/* float LBD (72-d, unit norm) Euclidean nearest neighbour on fp32 MFMA */
LF_API int lf_associate_float(lf_handle* h, const float* query72, int nq, const float* map72, int nm,
                       int32_t* idx, float* dist, int on_device);

/* The list forms of the matcher (SURVEY a-10): BinaryDescriptorMatcher::knnMatch (binary_descriptor_matcher.cpp:258-335)
 * and radiusMatch (:428-504) = Mihasher with K = k / K = N: the nearest map codes within D = 128 bits, nearest first.
 * Among equally near codes: the handle's tie rule (lf_set_tie_rule) -- LF_TIE_MIHASHER (default): the reference's discovery order,
 * i.e. (distance, discovery key, index) as Mihasher::query records them (:716-722); LF_TIE_LOWEST: (distance, index).
 * lf_knn_match    idx / dist [nq][k], k <= 16; slots beyond the matches within 128 bits: idx -1, dist -1 (the reference
 *                 leaves them unset)
 * lf_radius_match all map codes within min(max_distance, 128) bits per query as a CSR list: offsets [nq + 1], idx / dist
 *                 [cap] (distance ascending, then index); *total = number of matches; LF_ERR_CAPACITY when total > cap
 *                 (offsets and *total are complete then: size the arrays and call again)
 * Exact XOR / popcount, one lane per query (k_knn.hip); on_device applies to every array. */
LF_API int lf_knn_match(lf_handle* h, const uint8_t* query32, int nq, const uint8_t* map32, int nm, int k, int32_t* idx, float* dist,
                 int on_device);
/* The `mask` argument of match / knnMatch / radiusMatch (binary_descriptor_matcher.cpp:231-235, 305-309, 477-481): a DMatch is made
 * only for the queries whose mask byte is not 0, and carries its queryIdx.  lf_select_queries compacts those queries (in order)
 * into selected32 [<= nq][32] with their row numbers in query_idx; run any of the three forms on selected32 -- result row i
 * then belongs to query query_idx[i].  Blocking (returns the count). */
LF_API int lf_select_queries(lf_handle* h, const uint8_t* query32, int nq, const uint8_t* mask, uint8_t* selected32, int32_t* query_idx,
                      int* n_selected, int on_device);
LF_API int lf_radius_match(lf_handle* h, const uint8_t* query32, int nq, const uint8_t* map32, int nm, float max_distance,
                    int32_t* offsets, int32_t* idx, float* dist, int cap, int* total, int on_device);

/* BinaryDescriptorMatcher's DATASET form (binary_descriptor_matcher.cpp:70-111 add / train / clear, :117-195 match, :339-425
 * knnMatch, :508-595 radiusMatch): the descriptors of several train images are added one matrix at a time and searched as ONE set;
 * every DMatch says which image it came from.  As in the reference: trainIdx is the row number IN THE SET (not rebased to the
 * image), imgIdx the image whose rows hold it (indexesMap.upper_bound - 1, including std::map::insert's refusal to overwrite: an
 * image added right after an EMPTY one is reported under the empty one's number), and masks[imgIdx][query] == 0 drops a match
 * AFTER the search (it does not steer the search).  masks: NULL, or one pointer per added image (NULL = all ones) to nq bytes.
 * The searches are lf_associate / lf_knn_match / lf_radius_match on the set, with the handle's tie rule; host arrays; blocking.
 *   lf_matcher_match         out [<= nq]; *n_out matches (queries without a code within 128 bits make none)
 *   lf_matcher_knn_match     lists per query: list_offsets [nq + 1], out [<= nq k]; compact_result != 0 drops the empty lists
 *                            (*n_lists of them remain; without it list i is query i's)
 *   lf_matcher_radius_match  the same with every code within max_distance; *total = the matches the search found before the masks;
 *                            LF_ERR_CAPACITY when total > cap (size the arrays and call again) */
typedef struct { int32_t queryIdx, trainIdx, imgIdx; float distance; } lf_dmatch;
LF_API int lf_matcher_add(lf_handle* h, const uint8_t* codes32, int n, int on_device);
LF_API int lf_matcher_clear(lf_handle* h);
LF_API int lf_matcher_size(const lf_handle* h, int* n_images, int* n_descriptors);
LF_API int lf_matcher_match(lf_handle* h, const uint8_t* query32, int nq, const uint8_t* const* masks, lf_dmatch* out, int* n_out);
LF_API int lf_matcher_knn_match(lf_handle* h, const uint8_t* query32, int nq, int k, const uint8_t* const* masks, int compact_result,
                         int32_t* list_offsets, lf_dmatch* out, int* n_lists);
LF_API int lf_matcher_radius_match(lf_handle* h, const uint8_t* query32, int nq, float max_distance, const uint8_t* const* masks,
                            int compact_result, int32_t* list_offsets, lf_dmatch* out, int cap, int* n_lists, int* total);

/* Anti-instagram colour clustering (k-means part).  Replaces
 *   anti_instagram/kmeans.py:22-47  runKMeans(cv_img, num_colors, init)
 * = sklearn.cluster.KMeans(n_clusters, max_iter, init = <array>).fit_predict on B, G, R points + cluster_centers_, label
 * counts, score.  bgr_points: [n][3] u8 -- the reference passes the pixels of the frame's last 100 rows (any order: the
 * result does not depend on it except through which of several equally far samples re-seeds an empty cluster: the lowest
 * index).  init_centers [k][3] f64, k <= 16; max_iter 25 and tol 1e-4 are the reference's (scikit-learn's default tol).
 * centers_out [k][3] f64, counts_out [k], *inertia_out (score = -inertia), *n_iter_out.  Blocking.  LF_ERR_BAD_ARG when a
 * cluster stays empty (fewer distinct samples than clusters); LF_ERR_UNSUPPORTED for n > 2^24 points. */
LF_API int lf_kmeans(lf_handle* h, const uint8_t* bgr_points, int n, int on_device, int k, const double* init_centers, int max_iter,
              double tol, double* centers_out, long long* counts_out, double* inertia_out, int* n_iter_out);

/* ---- live map + associator (SURVEY a-11, 8f-3) ------------------------------------------------
 * What the package offers in place of the reference's line_associator node, which is an unfinished stub
 * (src/line_associator/src/line_associator_node.py:12-86), and of show_map's append-only segment store
 * (src/show_map/src/show_map.py:28-42).  The matching itself keeps BinaryDescriptorMatcher::match semantics
 * (a-10 above); everything else in this section is this package's OWN contract -- no reference behaviour exists
 * to match -- and oracle/lf_oracle_map.c is its sequential statement.
 *
 * A map lives on one device: per entry the 32-byte code, colour, the two ground endpoints (x0 y0 x1 y1, map
 * frame, metres), hits, last_seen (step number), plus the matrix-core operands of the associator (e2m1 nibbles and a colour row), which are
 * re-packed only for the rows an update touches.  All map work runs on the map's own HIP stream, in call order.
 *
 *   color_gating    0: a query may match any entry (a-10).  1: only entries of its own colour (Segment.color);
 *                   entries / queries with colour >= 3 match every colour.  Costs nothing: the test rides in
 *                   the matrix core (k_assoc.hip).
 *   max_distance    matches farther than this many bits are "no match" (<= 128; the reference's D = 128)
 *   kept_only       1: only segments line_sanity keeps (keep == 1) enter the map (show_map subscribes to the
 *                   filtered list, show_map_complete.launch:37)
 *   policy          LF_MAP_APPEND: every eligible segment is appended (show_map.py:41).
 *                   LF_MAP_MERGE: an eligible segment whose match is within merge_distance REFRESHES that
 *                   entry (code, colour, endpoints <- the segment's; hits += 1; last_seen = step; when several
 *                   segments of one update hit the same entry the last in SegmentList order wins); the others
 *                   are appended.
 *   when_full       LF_MAP_RING: appends wrap around and overwrite the oldest entries.
 *                   LF_MAP_FULL_ERROR: what does not fit is dropped and the failure is reported as below.
 * Failing updates (map full under LF_MAP_FULL_ERROR; a block with a bad header; a block carrying the overflow
 * marker) are detected on the device, so they are reported LATER and exactly ONCE: the first of lf_map_size,
 * lf_map_associate, lf_map_step, lf_map_step_host that sees the flag returns the error (LF_ERR_CAPACITY /
 * LF_ERR_BAD_ARG) WITHOUT doing its own work -- call it again.  Nothing is sticky: a full map can still be queried
 * and, under MERGE, refreshed.
 * One update consumes BLOCKS: [1 + rows][LF_BLOCK_ROW_BYTES] bytes each, row 0 = header
 * {u32 magic "LFBK", u32 count, i32 step, u32 n_frames, u32 overflow (0, or the segment count that did not fit:
 * count is 0 then), zeros}, then per segment, SegmentList order:
 *   0..31 code | 32..63 f64 x0 y0 x1 y1 in the MAP frame | 64 i32 idx | 68 f32 dist | 72 colour | 73 keep | 0-pad
 * (idx / dist = the segment's association result against the map as it stood BEFORE this update).  A block is
 * what ranks exchange with ONE all-gather per step; blocks are applied in the order given, so
 * replicas that see the same blocks hold the same map, and a single GPU applies its own block the same way.
 *
 * Map frame: odometry publishes map -> duck as (x, y, theta) (src/odometry/src/odometry.py:110-120);
 * lf_map_pack_block moves each segment's ground endpoints with ITS FRAME's pose:
 *   X = x + (cos(theta) * px - sin(theta) * py),  Y = y + (sin(theta) * px + cos(theta) * py)   (f64, unfused)
 * pose NULL = leave the points in the robot frame, as show_map.py does.
 */
typedef struct lf_map lf_map;
#define LF_MAP_APPEND 0
#define LF_MAP_MERGE 1
#define LF_MAP_RING 0
#define LF_MAP_FULL_ERROR 1
#define LF_BLOCK_ROW_BYTES 80
typedef struct lf_map_config {
    int32_t capacity;          /* entries, 64 .. 2^21 */
    int32_t color_gating;
    int32_t max_distance;      /* 0 .. 128 */
    int32_t policy;            /* LF_MAP_APPEND | LF_MAP_MERGE */
    int32_t kept_only;
    int32_t merge_distance;    /* MERGE: 0 .. max_distance */
    int32_t when_full;         /* LF_MAP_RING | LF_MAP_FULL_ERROR */
} lf_map_config;

LF_API int lf_map_create(int device_id, const lf_map_config* cfg, lf_map** out);
LF_API void lf_map_destroy(lf_map* m);
LF_API const char* lf_map_last_error(const lf_map* m);      /* m == NULL: the last lf_map_create failure */
LF_API int lf_map_get_stream(lf_map* m, void** hip_stream);
LF_API int lf_map_synchronize(lf_map* m);
/* append n entries as they are (color NULL: 255 = matches every colour; ground NULL: zeros); hits 1, last_seen -1 */
LF_API int lf_map_seed(lf_map* m, const uint8_t* code32, const uint8_t* color, const double* ground4, int n, int on_device);
/* entries in use, ring head, lifetime counters; waits for the map's stream; reports a failing update (once, the
 * outputs are still filled in) */
LF_API int lf_map_size(lf_map* m, int* size, int* head, int64_t* total_appended, int64_t* total_refreshed);
/* Nearest map entry of n queries (a-10 semantics + the map's gating / max_distance).  color may be NULL when
 * gating is off.  h: the handle whose stream produced the query arrays (the map's stream waits for it, and the
 * handle's next batch waits until the map has read them), or NULL when the caller has ordered that itself.
 * on_device applies to all four arrays; with host arrays the call returns when idx / dist are in place. */
LF_API int lf_map_associate(lf_map* m, lf_handle* h, const uint8_t* code32, const uint8_t* color, int n,
                     int32_t* idx, float* dist, int on_device);
/* tie rule of the map's associations (lf_map_associate, lf_map_step*): LF_TIE_MIHASHER unless set; with LF_TIE_MIHASHER and
 * colour gating the reference's discovery order applies among the entries the query may match */
LF_API int lf_map_set_tie_rule(lf_map* m, int tie_rule);
/* Device arrays of `segs` (frame_offset, code, color, keep, ground; capacity ignored) + idx / dist -> one block in
 * device memory.  block_rows < n + 1: LF_ERR_CAPACITY -- never truncated: the block is then ONLY a header with the
 * overflow marker, which a rank of a multi-GPU step still all-gathers so that every replica skips that step's update
 * together (lf_map_update) and reports LF_ERR_CAPACITY instead of waiting for a collective that never comes.
 * frame_pose: host [n_frames][3] = x, y, theta per frame, or NULL. */
LF_API int lf_map_pack_block(lf_map* m, lf_handle* h, const lf_segments* segs, int n, int n_frames, const int32_t* idx,
                      const float* dist, const double* frame_pose, int step, uint8_t* block, int block_rows);
/* apply n_blocks consecutive blocks of block_rows rows each (device memory), in order.  If any of them has a bad
 * header or the overflow marker, NONE is applied (reported as described above). */
LF_API int lf_map_update(lf_map* m, const uint8_t* blocks, int n_blocks, int block_rows);
/* single-GPU convenience: lf_map_associate + lf_map_pack_block + lf_map_update; idx / dist device arrays [n] */
LF_API int lf_map_step(lf_map* m, lf_handle* h, const lf_segments* segs, int n, int n_frames, const double* frame_pose,
                int step, int32_t* idx, float* dist);
/* lf_map_step with HOST arrays in `segs` (frame_offset, code, color, keep, ground as lf_process_batch returns them with
 * out_on_device = 0) and host idx / dist: upload, associate, update, download; returns when idx / dist are in place */
LF_API int lf_map_step_host(lf_map* m, const lf_segments* segs, int n, int n_frames, const double* frame_pose, int step,
                     int32_t* idx, float* dist);
/* copy entries [first, first + n) to host arrays (NULL arrays are skipped); waits for the map's stream */
LF_API int lf_map_fetch(lf_map* m, int first, int n, uint8_t* code32, uint8_t* color, double* ground4, int32_t* hits,
                 int32_t* last_seen);

/* per-stage timing with HIP events on the map's stream: 0 query packing (always 0 calls since the association became one
 * launch that expands the queries itself), 1 association (MFMA), 2 block packing,
 * 3 map update.  lf_map_get_timing returns what accumulated since the previous call and resets it. */
#define LF_MAP_N_STAGES 4
LF_API int lf_map_set_profiling(lf_map* m, int enabled);
LF_API int lf_map_get_timing(lf_map* m, double* ms_per_stage, int32_t* launches_per_stage, int n);
LF_API const char* lf_map_stage_name(int stage);

/* ---- EDLines detector + multi-octave KeyLines / LBD --------------------------------
 * The reference's second detector: BinaryDescriptor::operator() with useProvidedKeyLines = false
 * (src/line_descriptor/src/binary_descriptor_custom.cpp:263-301) = detectImpl (:455-513: OctaveKeyLines :689-1024,
 * one EDLineDetector per octave :1442-2751) followed by computeImpl on the DETECTOR's own gradient images
 * (:1079-1090).  Per octave: GaussianBlur(ksize 5, sigma 1, 1, sqrt 2, 2, ...), EDLines (Sobel, thresholded gradient,
 * anchors, smart routing, least-squares line fitting, NFA validation), resize by 1 / sqrt 2; the octaves' lines that
 * belong together share a class_id; KeyLines come ordered by class_id, then octave.  Parity: held bit for bit to
 * oracle/lf_oracle_edlines.c, which restates the in-tree C++ -- unpinned against a real build of it (needs OpenCV).
 *
 * lf_edlines_params   EDLineDetector::EDLineDetector() defaults (:1374-1385): gradient_threshold 80, anchor_threshold 8,
 *                     scan_intervals 2, min_line_len 15 (<= 64), line_fit_err_threshold 1.6; ksize 5 (odd, 1 .. 31)
 * lf_keylines         struct of arrays, caller allocated for `capacity` lines, NULL arrays are skipped; field meaning =
 *                     KeyLine (include/line_descriptor/descriptor_custom.hpp:105-144): start_end = startPointX/Y,
 *                     endPointX/Y (original image scale), in_octave = s/ePointInOctaveX/Y, angle = direction of the line
 *                     (dark side on its left), num_pixels, line_length, octave, class_id, response, size, pt (2 per
 *                     line), salience (OctaveSingleLine::salience, not a KeyLine field), desc (72 f32) / code (32 B)
 * input_kind          0: raw camera frames [n][in_rows][in_cols][3] BGR -- the gray image is BGR2GRAY of the handle's
 *                     working image (resize / crop / colour correction as in lf_process_batch); 1: gray working
 *                     images [n][img_rows - top_cutoff][img_cols] u8 as they are
 * frame_status        optional host [n_frames]: 0 ok; 1..4: the detector gave up on an octave of that frame (anchor /
 *                     edge arrays full, :1533-1537, :2185-2196; more than 5 lines per edge or than the handle holds;
 *                     more than 32 767 lines in the frame: the reference counts lines in a `short`) -- such a frame has no KeyLines, as in the reference, where
 *                     detectImpl ignores OctaveKeyLines' return value (:465-468)
 * Synchronous.  LF_ERR_CAPACITY when the KeyLines do not fit out->capacity. */
#define LF_MAX_OCTAVES 5
typedef struct lf_edlines_params {
    int32_t gradient_threshold, anchor_threshold, scan_intervals, min_line_len;
    double line_fit_err_threshold;
    int32_t ksize;
} lf_edlines_params;
typedef struct lf_keylines {
    int32_t capacity;
    int32_t* frame_offset;      /* n_frames + 1 */
    float* start_end;
    float* in_octave;
    float* angle;
    int32_t* num_pixels;
    float* line_length;
    int32_t* octave;
    int32_t* class_id;
    float* response;
    float* size;
    float* pt;
    float* salience;
    float* desc;
    uint8_t* code;
} lf_keylines;
LF_API void lf_edlines_default_params(lf_edlines_params* p);
/* BinaryDescriptor::Params and its setters (binary_descriptor_custom.cpp:108-200; descriptor_custom.hpp Params) on a handle:
 *   num_of_octave     numOfOctave_ (1): kept and returned; the entry points take n_octaves explicitly
 *   width_of_band     widthOfBand_ (7): setWidthOfBand (:134-176) -- the support region is 9 w rows, both Gaussian tables F_g (9 w) and
 *                     F_l (3 w) are recomputed with the reference's integer divisions.  Applies to EVERY descriptor the handle computes
 *                     from then on (lf_process_batch with describe, lf_keylines_batch, lf_lsd_keylines_batch, lf_describe_keylines).
 *                     1 .. 21, else LF_ERR_UNSUPPORTED
 *   reduction_ratio   reductionRatio (2): computeGaussianPyramid (:366) asks pyrDown for Size(cols / r, rows / r), which cv::pyrDown
 *                     only accepts within 2 pixels of half the source -- i.e. r = 2; with any other value a compute over more than one
 *                     octave fails there (cv::Exception) and here (LF_ERR_UNSUPPORTED from lf_describe_keylines / the describe step of
 *                     lf_lsd_keylines_batch); one octave never reaches the call
 *   ksize             ksize_ (5): the Gaussian of OctaveKeyLines (:708) when lf_keylines_batch is given no lf_edlines_params (a params
 *                     block names its own ksize); odd, 1 .. 31
 * Params::read / write (:189-204: a cv::FileStorage node with numOfOctave_, widthOfBand_, reductionRatio; write adds numOfBand_ = 9)
 * are host-side text handling: lane_slam_amd.matcher.BinaryDescriptorParams mirrors them.  Not while a batch is in flight. */
typedef struct {
    int32_t num_of_octave, width_of_band, reduction_ratio, ksize;
} lf_descriptor_params;
LF_API void lf_descriptor_default_params(lf_descriptor_params* p);
LF_API int lf_set_descriptor_params(lf_handle* h, const lf_descriptor_params* p);
LF_API int lf_get_descriptor_params(lf_handle* h, lf_descriptor_params* p);
LF_API int lf_keylines_batch(lf_handle* h, const uint8_t* images, int n_frames, int input_kind, int images_on_device, int n_octaves,
                      const lf_edlines_params* params_or_null, lf_keylines* out, int out_on_device, int describe,
                      int* n_keylines, int32_t* frame_status_or_null);
/* The `mask` argument of BinaryDescriptor::detect (binary_descriptor_custom.cpp:415-437, 509-519): masks = n_frames images of the
 * handle's WORKING size [rows][cols] u8; a KeyLine whose two end points (image coordinates, truncated) both lie on zero pixels is
 * erased -- BY THE REFERENCE'S LOOP AS WRITTEN, which does not step back after an erase: the KeyLine that slides into the erased
 * place is never tested, so of a run of consecutive KeyLines that fail the test the 1st, 3rd, 5th ... go and the 2nd, 4th ... stay.
 * (LSDDetectorC::detect has the step back: lf_lsd_keylines_batch_ex.)  out->capacity must hold the KeyLines BEFORE the mask. */
LF_API int lf_keylines_batch_masked(lf_handle* h, const uint8_t* images, int n_frames, int input_kind, int images_on_device, int n_octaves,
                             const lf_edlines_params* params_or_null, const uint8_t* masks, int masks_on_device, lf_keylines* out,
                             int out_on_device, int describe, int* n_keylines, int32_t* frame_status);
/* The pipelined form (like lf_process_batch_async): images and every out_dev array in device memory, the whole batch is queued
 * on the handle's stream and the call returns; lf_wait blocks and returns the KeyLine total in *n_segments (LF_ERR_CAPACITY
 * when it exceeds out_dev->capacity: no array is complete then); lf_keylines_frame_status copies the per-frame status of the
 * batch lf_wait completed.  One batch in flight per handle; several handles keep the chip busy while one frame's edge walk
 * is a single wave's chain. */
LF_API int lf_keylines_batch_async(lf_handle* h, const uint8_t* images_dev, int n_frames, int input_kind, int n_octaves,
                            const lf_edlines_params* params_or_null, lf_keylines* out_dev, int describe);
LF_API int lf_keylines_frame_status(lf_handle* h, int32_t* frame_status, int n_frames);
/* The library's OTHER detector: LSDDetectorC::detect (src/line_descriptor/src/LSDDetector_custom.cpp:49-72, 130-215) -- a gray
 * pyramid by pyrDown (scale 2, no blur), cv::createLineSegmentDetector() with its DEFAULT parameters (REFINE_STD) on every level,
 * one KeyLine per line (checkLineExtremes, start / end points scaled back to level 0, lineLength, numOfPixels = LineIterator's
 * count, angle, size, response, pt; class_id counts through the octaves of a frame; no mask) -- and with describe != 0
 * BinaryDescriptor::compute on those KeyLines (lf_describe_keylines).  Same lf_keylines block as lf_keylines_batch (salience is 0:
 * not a KeyLine field); same input kinds.  Synchronous.  Every level runs the front end's LSD kernels on a GRAY image through a
 * sub-handle of the level's geometry: dense problems, one wave per connected component -- a completeness path (detect -> compute
 * for both detectors of the library), not a fast one.  LF_ERR_CAPACITY: more KeyLines than out->capacity, or a level with more
 * lines than max_lines_per_color. */
LF_API int lf_lsd_keylines_batch(lf_handle* h, const uint8_t* images, int n_frames, int input_kind, int images_on_device, int n_octaves,
                          lf_keylines* out, int out_on_device, int describe, int* n_keylines);
/* The fork's own overloads of the same detector (the reason its files are called _custom): LSDDetectorC::detect(image, keylines,
 * scale, numOctaves, LSDOptions, mask) and detectFast -- LSDDetector_custom.cpp:218-325 and :327-438 are the same text -- i.e. the
 * caller's parameters for cv::createLineSegmentDetector (descriptor_custom.hpp:906-916), `length > opts.min_length` (:277; class_id
 * counts the kept lines), and the mask argument (:203-213, :312-322): KeyLines whose two end points (image coordinates, truncated)
 * BOTH lie on zero mask pixels are erased.
 *   opts   NULL = lf_lsd_keylines_batch (OpenCV's defaults, no length test)
 *   masks  NULL, or n_frames images of the handle's WORKING size [rows][cols] u8
 * With lsd_seed_order = LF_LSD_SEED_OPENCV32: (n_bins - 1) * quant / sin(ang_th) >= 361 (a pixel with a defined gradient must not
 * fall into bin 0), else LF_ERR_UNSUPPORTED. */
typedef struct {
    int32_t refine;              /* cv::LSD_REFINE_NONE 0, _STD 1, _ADV 2 */
    int32_t n_bins;
    double scale, sigma_scale, quant, ang_th, log_eps, density_th;
    double min_length;
} lf_lsd_options;
LF_API void lf_lsd_default_options(lf_lsd_options* opts);      /* createLineSegmentDetector()'s: 1, 1024, 0.8, 0.6, 2.0, 22.5, 0, 0.7; min_length 0 */
LF_API int lf_lsd_keylines_batch_ex(lf_handle* h, const uint8_t* images, int n_frames, int input_kind, int images_on_device, int n_octaves,
                             const lf_lsd_options* opts, const uint8_t* masks, int masks_on_device,
                             lf_keylines* out, int out_on_device, int describe, int* n_keylines);
/* Plugin path with the EDLines detector: lf_set_image_edlines, then lf_detect_lines exactly as after lf_set_image.
 * The reference has ONE LineDetectorInterface implementation working on colour masks (LineDetectorLSD,
 * line_detector_lsd.py:11-142); this is the package's second (SURVEY 8f-4 "alternative detector plugin") and its
 * contract is the package's own: EDLines (one octave; KeyLine endpoints sPointInOctave -> ePointInOctave) runs on
 * BGR2GRAY of the working image, a line belongs to colour c when the dilated colour mask `bw` of that colour
 * (line_detector_lsd.py:38-58) is set under the truncated, clamped centre of the line; normals, centres and the
 * endpoint ordering come from the same code as for LSD lines (_findNormal / _correctPixelOrdering, :74-125).
 * LF_ERR_CAPACITY when the detector gives up on the image (the reference prints "Line Detection not finished"). */
LF_API int lf_set_image_edlines(lf_handle* h, const uint8_t* bgr, int rows, int cols, int row_stride_bytes,
                         const lf_edlines_params* params_or_null);
/* BinaryDescriptor::compute on GIVEN KeyLines (:524-687, useDetectionData = false): gradients from
 * computeGaussianPyramid (:350-371: GaussianBlur 5x5 sigma 1, then pyrDown by 2 per octave) + Sobel (:374-398).
 * gray: [n_frames][rows][cols] u8 working images; per line: its frame, in_octave endpoints (4), angle, num_pixels,
 * octave (< LF_MAX_OCTAVES).  desc [n][72] / code [n][32], either may be NULL.  All arrays host (on_device = 0) or
 * device.  The multi-octave LSD KeyLines of LSDDetector_custom.cpp:130-215 (scale 2) are what lives on this pyramid. */
LF_API int lf_describe_keylines(lf_handle* h, const uint8_t* gray, int n_frames, const int32_t* line_frame, const float* in_octave4,
                         const float* angle, const int32_t* num_pixels, const int32_t* octave, int n, float* desc72,
                         uint8_t* code32, int on_device);
/* intermediate results of the last lf_keylines_batch for tests (synchronises): `what` = 0 blurred octave image (u8)
 * 1 dx | dy << 16 (u32) 2 thresholded gradient / 4 | direction << 15 (u16) 3 anchors (u32 x | y << 16, [frames][cap])
 * 4 edge chains (u32, [frames][2 cap]) 5 chain starts (u32, [frames][max_edges + 2]) 6 counts (i32 [frames][4]: anchors,
 * edges, lines, status) 7 line endpoints (f32 x4, [frames][max_lines]) 8 lineEquation[2] (f64) 9 direction (f32)
 * 10 pixels per line (i32) 11 salience (f32) 12 the octave's input image (u8); dims: octave rows, cols, cap, max_edges,
 * max_lines */
LF_API int lf_keylines_debug_fetch(lf_handle* h, int octave, int what, void* dst, size_t bytes, int32_t* dims5_or_null);

/* ---- host ingest: replaces duckietown_utils.jpg.image_cv_from_jpg ---------------
 * = cv2.imdecode(np.fromstring(data, np.uint8), cv2.IMREAD_COLOR)
 * (src/duckietown/include/duckietown_utils/jpg.py:21-31, called per frame from
 *  src/line_detector/src/line_detector_node.py:153-158), i.e. libjpeg-turbo's default decoder.
 * Host threads parse and Huffman-decode the streams into sparse coefficient lists; dequantisation,
 * inverse DCT, chroma upsampling and YCbCr -> BGR run on the handle's stream.  Output: u8 BGR
 * [n_frames][rows][cols][3], bit identical to libjpeg-turbo for baseline / extended-sequential
 * Huffman streams with 4:4:4, 4:2:2, 4:2:0 or grayscale sampling (anything else: LF_ERR_UNSUPPORTED).
 *
 * jpeg[i] / jpeg_size[i]  host pointers to the n_frames streams
 * rows, cols              expected image size; a stream of another size gets LF_ERR_BAD_ARG
 * frames                  device pointer (frames_on_device = 1: the call returns once the work is
 *                         queued on the handle's stream, order it with lf_synchronize or simply pass the
 *                         buffer to lf_process_batch on the same handle) or host pointer (= 0: synchronous)
 * n_threads               host threads for entropy decoding (<= 0: one per frame, at most 64)
 * frame_status            optional [n_frames] lf_status per frame.  A frame that cannot be decoded is
 *                         written as zeros -- the reference logs and drops such a frame
 *                         (line_detector_node.py:155-158).  Without frame_status the call returns
 *                         LF_ERR_DECODE if any frame failed.
 */
LF_API int lf_jpeg_decode_batch(lf_handle* h, const uint8_t* const* jpeg, const size_t* jpeg_size, int n_frames,
                         int rows, int cols, uint8_t* frames, int frames_on_device, int n_threads,
                         int* frame_status);
/* The same with the ENTROPY DECODER ON THE DEVICE as well (k_jhuff.hip): the host only parses the headers; unstuffing,
 * Huffman decoding (self-synchronising subsequences of 48 bytes, decoded once per place in the MCU), DC prediction and everything after it run on
 * the handle's stream.  Same streams accepted, same output bits, same per-frame status as lf_jpeg_decode_batch.  n_threads:
 * host threads for header parsing and for copying the entropy-coded bytes into pinned memory (<= 0: up to 16).  The call
 * returns when the batch is decoded (the per-frame status comes from the device). */
LF_API int lf_jpeg_decode_batch_gpu(lf_handle* h, const uint8_t* const* jpeg, const size_t* jpeg_size, int n_frames,
                             int rows, int cols, uint8_t* frames, int frames_on_device, int n_threads,
                             int* frame_status);
/* The same, QUEUED (round 6): frames_device is a device address (the handle's own buffer, lf_frames_buffer, or the caller's); the call
 * returns when the headers are parsed and the work is on the handle's stream -- hand the buffer to lf_process_batch_async on the
 * same handle next, nothing in between waits for the device (lf_jpeg_decode_batch_gpu returns when the batch is decoded: a feeder
 * thread spent the decoder's 1 - 3 ms per batch inside it).  The per-frame status (as lf_jpeg_decode_batch's frame_status; frames
 * that could not be decoded are zeros) is read with lf_jpeg_status, which waits for the decode of the handle's last queued batch
 * alone; one queued batch per handle at a time.  n_failed: optional count of frames whose status is not LF_OK. */
LF_API int lf_jpeg_decode_batch_gpu_async(lf_handle* h, const uint8_t* const* jpeg, const size_t* jpeg_size, int n_frames,
                                   int rows, int cols, uint8_t* frames_device, int n_threads);
LF_API int lf_jpeg_status(lf_handle* h, int* frame_status, int n_frames, int* n_failed);
/* Queued decode of what lf_process_batch ON THIS HANDLE will read, into the handle's own frame buffer (lf_frames_buffer): streams of
 * the configured input size; of every frame only the rows from the crop line on are produced (top_cutoff after the resize: a third of
 * a 640 x 480 camera frame is never looked at by line_detector_node.py:163-166's crop, so its inverse DCT, upsampling and colour
 * conversion are skipped) -- the rows above keep whatever the buffer held.  Follow with lf_process_batch_async(h, buffer, n, 1, ...);
 * status through lf_jpeg_status as for lf_jpeg_decode_batch_gpu_async. */
LF_API int lf_jpeg_decode_for_detect_async(lf_handle* h, const uint8_t* const* jpeg, const size_t* jpeg_size, int n_frames, int n_threads);
/* size and layout of one stream without decoding it (hmax x vmax = luma sampling factors) */
LF_API int lf_jpeg_info(const uint8_t* jpeg, size_t jpeg_size, int* rows, int* cols, int* components, int* hmax, int* vmax);
/* the handle's own device staging buffer for input frames ([max_frames][in_rows][in_cols][3] u8): decode
 * into it, then hand the same pointer to lf_process_batch with frames_on_device = 1 */
LF_API int lf_frames_buffer(lf_handle* h, uint8_t** device_ptr, size_t* bytes);

/* ---- SegmentList glue ---------------------------------------------------------------
 * The reference hands segments from node to node as duckietown_msgs/SegmentList and builds / walks them one
 * Python object at a time (line_detector_node.py:251-265 toSegmentMsg, ground_projection_node.py:55-65,
 * line_sanity_node.py:48-72).  These two calls convert between the struct-of-arrays block and the ROS 1 wire
 * form of `duckietown_msgs/Segment[] segments` (src/duckietown_msgs/msg/Segment.msg:1-8, Vector2D.msg:1-2,
 * geometry_msgs/Point): little endian, u32 count, then 73 bytes per segment
 *   u8 color | f32 pixels_normalized[0].x .y [1].x .y | f32 normal.x .y | f64 points[0].x .y .z [1].x .y .z
 * so a node publishes  serialised Header + body  without touching a segment in Python.
 *   LF_MSG_DETECTOR  what line_detector_node publishes: color, pixels_normalized, normal (points 0)
 *   LF_MSG_GROUND    what ground_projection_node publishes: color, points with z = 0 (the rest 0)
 *   LF_MSG_FILTERED  what line_sanity_node publishes: the LF_MSG_GROUND segments with keep == 1, order kept
 * lf_serialize_segments: segs needs frame_offset, color and the stage's arrays (host or device pointers, all
 *   the same kind); out receives the n_frames bodies back to back, frame_byte_offset[n_frames + 1] (host) where
 *   each starts.  LF_ERR_CAPACITY if out_capacity is too small (frame_byte_offset[n_frames] then holds the need).
 * lf_deserialize_segments: the inverse, every field of the message is kept (frame_offset, color,
 *   pixels_normalized, normals, ground x/y of the two points; NULL arrays are skipped); LF_ERR_DECODE if a
 *   body's count does not match its length.
 */
#define LF_MSG_DETECTOR 0
#define LF_MSG_GROUND 1
#define LF_MSG_FILTERED 2
LF_API int lf_serialize_segments(lf_handle* h, const lf_segments* segs, int segs_on_device, int n_frames, int stage,
                          uint8_t* out, size_t out_capacity, int out_on_device, int64_t* frame_byte_offset);
LF_API int lf_deserialize_segments(lf_handle* h, const uint8_t* bodies, int bodies_on_device, const int64_t* frame_byte_offset,
                            int n_frames, lf_segments* out, int out_on_device, int* n_segments);

/* ---- introspection for tests and the benchmark ---------------------------- */
typedef enum lf_buffer_id {
    LF_BUF_BGR = 0,          /* u8  [frames][Hc][W][3]   corrected working image          */
    LF_BUF_MASKS = 1,        /* u8  [frames][3][Hc][W]   dilated colour masks 0/255       */
    LF_BUF_EDGES = 2,        /* u8  [frames][Hc][W]      Canny edges 0/255                */
    LF_BUF_LSD_ANGLE = 3,    /* f32 [frames][3][Hs][Ws]  level-line angle, degrees, NOTDEF = -1024 (rebuilt from the compact arrays) */
    LF_BUF_LSD_MODGRAD = 4,  /* f64 [frames][3][Hs][Ws]  gradient magnitude where defined, 0 elsewhere */
    LF_BUF_LSD_ORDER = 5,    /* u32 [frames][3][Hs*Ws]   seed order: (n_bins-1-bin) << 20 | compact index */
    LF_BUF_LSD_NORDER = 6,   /* i32 [frames][3]          seeds per run                    */
    LF_BUF_LBD_DX = 7,       /* i16 [frames][Hc][W]                                        */
    LF_BUF_LBD_DY = 8,       /* i16 [frames][Hc][W]                                        */
    LF_BUF_LSD_COUNTS = 9,   /* i32 [frames][3]          lines per run                    */
    LF_BUF_LSD_SCRATCH = 10, /* u32 [frames][3][Hs*Ws]   region-list scratch (diagnostic builds park counters here) */
    LF_BUF_LSD_NLOW = 11     /* i32 [frames][3]   pixels with a non-zero gradient below the threshold (the low records of lsd_seed_order = OPENCV32; 0 otherwise) */
} lf_buffer_id;
/* copy an intermediate buffer of the last batch to host memory (synchronises) */
LF_API int lf_debug_fetch(lf_handle* h, int buffer_id, void* dst, size_t bytes);
/* evaluate one deterministic-math routine on the device (host arrays in/out):
 * which = 0 exp 1 log 2 sin 3 cos 4 atan 5 asin 6 log10 7 sinh_small 8 atan2(a,b) 9 pow(a,b)
 *         10 sqrt 11 a/b 12 fastAtan2(float a, float b) 13 sqrtf 14 float a/b */
LF_API int lf_debug_detmath(lf_handle* h, int which, const double* a, const double* b_or_null, double* y, int n);
/* counter calibration: stream `bytes` of a scratch buffer `reps` times with `width` (4 / 8 / 12 / 16) bytes per lane
 * per access (write != 0: stores, 4 or 16); run under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` (tools/fetch_probe.py) */
LF_API int lf_debug_probe(lf_handle* h, int width, int write, size_t bytes, int reps);
/* LSD stages alone on a binary image of the handle's working size (non-zero = edge pixel, colour
 * mask forced to all ones); host pointers; lines before normal-based endpoint ordering */
LF_API int lf_debug_lsd_binary(lf_handle* h, const uint8_t* img, int rows, int cols, float* lines4, int cap, int* n_out);
/* the sort emulation behind lsd_seed_order = LF_LSD_SEED_OPENCV32 alone: order[i] = index of the element that
 * std::sort(begin, end, [](a, b) { return a.key > b.key; }) of libstdc++ leaves at place i, for n keys in [0, 1023] in their
 * initial order (host pointers; n < 2^20).  Elements with key 0 are the detector's flat pixels -- never seeds, anonymous on the
 * device (the sparse form keeps only the non-zero keys): the elements with a non-zero key come first, in std::sort's order, the
 * zero-key ones follow by index (std::sort leaves them behind the others too, in an order nothing observes) */
LF_API int lf_debug_std_sort(lf_handle* h, const int32_t* keys, int n, int32_t* order);
/* scaled LSD image size for this handle */
LF_API int lf_lsd_size(const lf_handle* h, int* rows, int* cols);
/* How many batches (handles) a caller should keep in flight for the content this handle saw last: region growing is a chain
 * of dependent steps per problem, and on busy content (camera frames with texture: a few problems of 10 - 20 k edge pixels set
 * the batch's latency while most of the chip waits) only more batches in flight fill the machine.  8 (6 until the kernels of round 4: six and eight measured 144 - 146 k and 147 - 148 k frames/s) while the last batch's
 * problems fit the small LDS slice of k_lsd_grow (lane markings), 18 otherwise (measured on camera frames, round 4: 6 / 12 / 18 in
 * flight = 50 k / 56 k / 62 k frames/s; give the HIP runtime more hardware queues than that: GPU_MAX_HW_QUEUES, INTEGRATION.md
 * section 4).  A hint: results never depend on it. */
LF_API int lf_suggested_depth(const lf_handle* h);

/* The per-problem lists of the LSD stages (records of defined pixels, compact arrays, seed lists, sort scratch: ~100 bytes per entry
 * and (frame, colour)).  A handle for more than 16 frames starts with an eighth of the LSD image per problem -- a lane frame's colour has
 * 3 - 6 % of its pixels defined, a camera frame's 10 - 20 % -- and when a batch holds a problem with more, lf_wait reallocates the lists
 * with room to spare and runs that batch again (results never depend on the capacity; only that one batch takes twice as long).
 * *entries = the current capacity per problem, *grown = how many times it was raised.  LF_LSD_RECORDS=<entries> | full in the
 * environment sets the starting capacity of handles created afterwards (full = the whole LSD image: never a second run). */
LF_API int lf_lsd_list_capacity(const lf_handle* h, int* entries, int* grown);
/* u32 words per (frame, colour) of LF_BUF_LSD_SCRATCH (lf_debug_fetch): follows the list capacity */
LF_API int lf_lsd_scratch_stride(const lf_handle* h);

/* per-kernel timing with HIP events on the handle's stream */
#define LF_N_STAGES 14
LF_API int lf_set_profiling(lf_handle* h, int enabled);
/* ms accumulated per stage since the last reset, and launches counted */
LF_API int lf_get_timing(lf_handle* h, double* ms_per_stage, int32_t* launches_per_stage, int n);
LF_API int lf_reset_timing(lf_handle* h);
LF_API const char* lf_stage_name(int stage);

#ifdef __cplusplus
}
#endif
#endif /* LANEFRONT_H */
