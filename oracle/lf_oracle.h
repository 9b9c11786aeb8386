/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded CPU restatement of the lane-slam line-feature front
 * end (detect -> describe -> ground-project -> sanity -> associate).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (lane_slam_amd/, liblanefront.so) never does.
 *
 * PARITY STATUS (see DESIGN.md "Oracle pinning"):
 *   pinned by the reference's own runnable Python (tests/golden/ npz fixtures):
 *     scaleandshift2, _findNormal/_checkBounds/_correctPixelOrdering,
 *     fancyFilters/processSegmentList.
 *   restated from in-tree reference source, PARITY UNPINNED (C++ needs OpenCV
 *   headers to build, none in this image): LBD (computeLBD, binaryConversion),
 *   Hamming matcher semantics, vector2pixel/pixel2ground arithmetic.
 *   restated from the published OpenCV 3.x algorithms the reference calls,
 *   PARITY UNPINNED (OpenCV is not vendored and not installed): resize-nearest,
 *   convertScaleAbs, BGR2HSV, BGR2GRAY, inRange, dilate, Canny, LSD,
 *   GaussianBlur, Sobel, undistortPoints, LineIterator count.
 *
 * Every function cites the reference file:line it follows.
 */
#ifndef LF_ORACLE_H
#define LF_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LFO_WHITE 0
#define LFO_YELLOW 1
#define LFO_RED 2

typedef struct lfo_config {
    /* geometry: line_detector_node.py:163-169, default.yaml:1-2 */
    int32_t in_rows, in_cols;        /* incoming frame, e.g. 480 x 640 */
    int32_t img_rows, img_cols;      /* img_size */
    int32_t top_cutoff;
    /* AntiInstagram transform, channel order = image channel order (BGR): scale_and_shift.py:25-33 */
    float ai_scale[3], ai_shift[3];
    /* HSV boxes: [0]=white, [1]=yellow, [2]=red1..red2, [3]=red3..red4 (line_detector_lsd.py:38-47) */
    int32_t hsv_lo[4][3], hsv_hi[4][3];
    int32_t dilation_kernel_size;    /* line_detector_lsd.py:52 */
    double canny_lo, canny_hi;       /* line_detector_lsd.py:61 */
    /* cv2.createLineSegmentDetector defaults (line_detector_lsd.py:65) */
    int32_t lsd_refine;              /* 0 none, 1 std, 2 adv */
    int32_t lsd_n_bins;
    double lsd_scale, lsd_sigma_scale, lsd_quant, lsd_ang_th, lsd_log_eps, lsd_density_th;
    /* ground projection: GroundProjection.py:38-78, calibrations yaml */
    double H[9], K[9], D[5], R[9], P[12];
    int32_t cam_w, cam_h;
    /* line_sanity_node.py:17-23 */
    double lanewidth, linewidth_white, linewidth_yellow, d_min, d_max, phi_min, phi_max;
    /* 0: seeds in raster order inside a bin (OpenCV 3.0 / 3.1); 1: std::sort's order (3.2+, lf_oracle_sort.cpp) */
    int32_t lsd_seed_order;
    int32_t reserved0;
} lfo_config;

/* working image size after resize+crop */
static inline int lfo_work_rows(const lfo_config* c) { return c->img_rows - c->top_cutoff; }
static inline int lfo_work_cols(const lfo_config* c) { return c->img_cols; }

/* ---- image stages (lf_oracle_image.c) ---- */
/* a-1: resize-nearest + crop + scaleandshift2 + convertScaleAbs. out: Hc*W*3 u8 */
void lfo_preprocess(const lfo_config* c, const uint8_t* bgr_in, uint8_t* bgr_out);
/* a-2: cvtColor BGR2HSV 8-bit */
void lfo_bgr2hsv(const uint8_t* bgr, int npix, uint8_t* hsv);
/* a-3: inRange masks (red = OR of two boxes); bw: 3 planes of npix */
void lfo_color_masks(const lfo_config* c, const uint8_t* hsv, int npix, uint8_t* bw3);
/* a-3: dilate with MORPH_ELLIPSE ksize x ksize */
void lfo_dilate_ellipse(const uint8_t* src, int rows, int cols, int ksize, uint8_t* dst);
/* a-2: Canny on 3-channel u8, aperture 3, L1 norm */
void lfo_canny_bgr(const uint8_t* bgr, int rows, int cols, double lo, double hi, uint8_t* edges);
/* a-9 inputs: BGR2GRAY, GaussianBlur 5x5 sigma 1 (u8), Sobel 3x3 -> s16 */
void lfo_bgr2gray(const uint8_t* bgr, int npix, uint8_t* gray);
void lfo_gaussian5_u8(const uint8_t* src, int rows, int cols, uint8_t* dst);
void lfo_sobel3_s16(const uint8_t* src, int rows, int cols, int16_t* dx, int16_t* dy);

/* ---- LSD (lf_oracle_lsd.c) ---- */
/* scaled-image size for an input of rows x cols */
void lfo_lsd_scaled_size(const lfo_config* c, int rows, int cols, int* srows, int* scols);
/* Gaussian + resize (f64) */
void lfo_lsd_scaled_image(const lfo_config* c, const uint8_t* img, int rows, int cols, double* scaled);
/* ll_angle: angle (rad, NOTDEF=-1024), modgrad, and the pseudo-ordered seed list
   (pixel addresses, descending bin then raster); returns list length */
int lfo_lsd_ll_angle(const lfo_config* c, const double* scaled, int srows, int scols,
                     double* angles, double* modgrad, int32_t* order);
/* seed order inside a gradient bin: 0 raster (default), 1 libstdc++ std::sort as the later OpenCV 3.x (lf_oracle_sort.cpp) */
void lfo_lsd_set_seed_order(int mode);
/* test helpers (lf_oracle_sort.cpp): the real std::sort(compare_norm) on a key array (returns its comparison count), and a
 * killer input for it (McIlroy's adversary) that drives it into the heap-sort fallback */
long long lfo_std_sort_keys(const int32_t* keys, int n, int32_t* order);
void lfo_antiqsort_keys(int n, int32_t* keys);
/* full detector: lines (x1,y1,x2,y2) float32, returns count (<= cap); extra = width,prec,nfa per line or NULL */
int lfo_lsd_detect(const lfo_config* c, const uint8_t* img, int rows, int cols,
                   float* lines4, double* extra3, int cap);

/* ---- per-segment stages (lf_oracle_segments.c) ---- */
/* a-5: normals (f64, value-equal to the f32 product), centers f32, lines reordered in place */
void lfo_find_normals(const uint8_t* bw, int rows, int cols, float* lines4, int n,
                      double* normals2, float* centers2);
/* a-6: float32((float64(x)+cut) * (1/size)) */
void lfo_normalize_lines(const lfo_config* c, const float* lines4, int n, float* out4);
/* a-7: vector2ground for both endpoints; pts: n*4 doubles (x0,y0,x1,y1), z==0 */
void lfo_ground_project(const lfo_config* c, const float* pixels_normalized4, int n, double* pts4);
/* a-8: keep mask + (d_i, phi_i, l_i, state) per segment */
void lfo_line_sanity(const lfo_config* c, const double* pts4, const uint8_t* color, int n,
                     uint8_t* keep, double* dphil3, int32_t* state);

/* ---- LBD + matcher (lf_oracle_lbd.c) ---- */
/* KeyLine fields used by computeLBD, from LSD lines in octave 0 (LSDDetector_custom.cpp:169-197) */
void lfo_keylines(const float* lines4, int n, int rows, int cols,
                  float* ext4 /*clamped sx,sy,ex,ey*/, float* angle, int32_t* num_pixels);
/* a-9: float (72) and binary (32 B) LBD descriptors */
void lfo_lbd(const int16_t* dx, const int16_t* dy, int rows, int cols,
             const float* ext4, const float* angle, const int32_t* num_pixels, int n,
             float* desc72, uint8_t* code32);
/* BinaryDescriptor::Params::widthOfBand_ for every later lfo_lbd call (default 7, 1 .. 32) */
void lfo_lbd_set_width_of_band(int w);
int lfo_lbd_get_width_of_band(void);
/* a-10: exact Hamming NN, distance > 128 => idx -1, dist -1; ties -> lowest train index */
void lfo_match(const uint8_t* query32, int nq, const uint8_t* train32, int nt,
               int32_t* idx, float* dist);
/* the same with the reference's tie rule (first discovered by Mihasher::query) and the number of equally near codes */
void lfo_match_mih(const uint8_t* query32, int nq, const uint8_t* train32, int nt, int32_t* idx, float* dist, int32_t* n_ties);
long long lfo_mih_discovery_key(const uint8_t* query32, const uint8_t* train32);
/* knnMatch / radiusMatch with the reference's order among equally near codes: (distance, discovery key, index) */
void lfo_knn_match_mih(const uint8_t* q, int nq, const uint8_t* t, int nt, int k, int32_t* idx, float* dist);
int lfo_radius_match_mih(const uint8_t* q, int nq, const uint8_t* t, int nt, float max_distance, int32_t* offsets, int32_t* idx, float* dist);
void lfo_knn_match(const uint8_t* query32, int nq, const uint8_t* train32, int nt, int k, int32_t* idx, float* dist);
int lfo_radius_match(const uint8_t* query32, int nq, const uint8_t* train32, int nt, float max_distance, int32_t* offsets, int32_t* idx,
                     float* dist);
/* float-descriptor L2 NN (72-d) */
void lfo_match_float(const float* q72, int nq, const float* t72, int nt, int32_t* idx, float* dist);

/* ---- EDLines + multi-octave KeyLines / LBD (lf_oracle_edlines.c) ---- */
typedef struct lfo_edlines_params {      /* EDLineDetector::EDLineDetector(), binary_descriptor_custom.cpp:1374-1385 */
    int32_t ksize; float sigma;          /* unused by the detector itself (the caller blurs), kept for the record */
    int32_t gradient_threshold, anchor_threshold, scan_intervals, min_line_len;
    double line_fit_err_threshold;
} lfo_edlines_params;
void lfo_edlines_params_default(lfo_edlines_params* p);
void lfo_gaussian_taps_q8(int ksize, double sigma, int32_t* taps);
void lfo_gaussian_blur_u8(const uint8_t* src, int rows, int cols, int ksize, double sigma, uint8_t* dst);
void lfo_resize_size(int rows, int cols, double inv_scale, int* drows, int* dcols);
void lfo_resize_linear_u8(const uint8_t* src, int rows, int cols, double inv_scale, uint8_t* dst);
void lfo_pyrdown_u8(const uint8_t* src, int rows, int cols, uint8_t* dst);
void lfo_ed_gradient(const uint8_t* img, int rows, int cols, int gradient_threshold, int16_t* dx, int16_t* dy, int16_t* g,
                     int16_t* gwo, uint8_t* dir);
int lfo_ed_anchors(const int16_t* g, const uint8_t* dir, int rows, int cols, int anchor_threshold, int scan, uint32_t* ax,
                   uint32_t* ay, int cap);
int lfo_ed_link(const int16_t* g, const uint8_t* dir, int rows, int cols, const uint32_t* ax, const uint32_t* ay, int n_anchors,
                int min_line_len, uint32_t* xcors, uint32_t* ycors, uint32_t* sid, uint8_t* edge_out);
double lfo_ed_nfa(int n, int k, double p, double logNT);
int lfo_ed_lines(const int16_t* pdx, const int16_t* pdy, const uint8_t* dir, int rows, int cols, const uint32_t* xcors,
                 const uint32_t* ycors, const uint32_t* sid, int n_edges, int min_line_len, double fit_err_threshold,
                 uint32_t* lx, uint32_t* ly, uint32_t* lsid, double* equations3, float* endpoints4, float* direction, int cap_lines);
void lfo_ed_salience(const int16_t* gwo, int cols, const uint32_t* lx, const uint32_t* ly, const uint32_t* lsid, int n_lines,
                     float* salience);
/* one octave, all stages kept: which = 0 dx 1 dy 2 g 3 gwo (s16) | 4 dir 5 edge (u8) | 6 ax 7 ay 8 xcors 9 ycors 10 sid
 * 11 lx 12 ly 13 lsid (u32) | 14 equations (f64 x3) | 15 endpoints (f32 x4) 16 direction 17 salience (f32) */
typedef struct lfo_edlines lfo_edlines;
lfo_edlines* lfo_edlines_run(const lfo_edlines_params* p, const uint8_t* blurred, int rows, int cols);
void lfo_edlines_free(lfo_edlines* e);
int lfo_edlines_counts(const lfo_edlines* e, int* n_anchors, int* n_edges, int* n_edge_pixels, int* n_lines, int* n_line_pixels);
const void* lfo_edlines_array(const lfo_edlines* e, int which);
/* KeyLines of BinaryDescriptor::detect + their descriptors (operator(), :263-301), one row per KeyLine */
typedef struct lfo_keylines_out {
    float* start_end;      /* startPointX, startPointY, endPointX, endPointY (original image scale) */
    float* in_octave;      /* sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY */
    float* angle;          /* KeyLine::angle = the line's direction, dark side on the left */
    int32_t* num_pixels;
    float* line_length;
    int32_t* octave;
    int32_t* class_id;
    float* response;       /* may be NULL, like the next four */
    float* size;
    float* pt;             /* 2 per line */
    float* salience;       /* OctaveSingleLine::salience (not part of KeyLine) */
    float* desc;           /* 72 per line */
    uint8_t* code;         /* 32 per line */
    int32_t octave_rows[8], octave_cols[8], octave_lines[8];
} lfo_keylines_out;
int lfo_octave_keylines(const lfo_edlines_params* p, const uint8_t* gray, int rows, int cols, int n_octaves, int ksize, int cap,
                        lfo_keylines_out* out);
int lfo_describe_keylines(const uint8_t* gray, int rows, int cols, const float* in_octave4, const float* angle,
                          const int32_t* num_pixels, const int32_t* octave, int n, float* desc72, uint8_t* code32);

/* ---- live map + associator (lf_oracle_map.c): the build's own contract, see that file's header ---- */
#define LFO_MAP_APPEND 0
#define LFO_MAP_MERGE 1
#define LFO_MAP_RING 0
#define LFO_MAP_FULL_ERROR 1
typedef struct lfo_map_config {
    int32_t capacity, color_gating, max_distance, policy, kept_only, merge_distance, when_full;
} lfo_map_config;
typedef struct lfo_map lfo_map;
lfo_map* lfo_map_create(const lfo_map_config* cfg);
void lfo_map_destroy(lfo_map* m);
void lfo_map_state(const lfo_map* m, int32_t* size, int32_t* head, int32_t* overflow, long long* total_appended,
                   long long* total_refreshed);
const uint8_t* lfo_map_codes(const lfo_map* m);
const uint8_t* lfo_map_colors(const lfo_map* m);
const double* lfo_map_ground(const lfo_map* m);
const int32_t* lfo_map_hits(const lfo_map* m);
const int32_t* lfo_map_last_seen(const lfo_map* m);
void lfo_map_seed(lfo_map* m, const uint8_t* code32, const uint8_t* color, const double* ground4, int n);
/* 0 (default): ties -> lowest index; 1: the reference's rule, the candidate Mihasher::query discovers first (lfo_match_mih) */
void lfo_map_set_tie_rule(lfo_map* m, int rule);
void lfo_map_associate(const lfo_map* m, const uint8_t* code32, const uint8_t* color, int n, int32_t* idx, float* dist);
void lfo_map_to_map_frame(const double* ground4, int n, const int32_t* frame_offset, int n_frames, const double* pose3,
                          double* out4);
void lfo_map_update(lfo_map* m, const uint8_t* code32, const uint8_t* color, const uint8_t* keep, const double* ground4,
                    const int32_t* idx, const float* dist, int n, int step);

/* ---- whole frame (lf_oracle_frame.c) ---- */
typedef struct lfo_frame_out {
    int32_t n;                  /* segments, order white, yellow, red */
    int32_t n_color[3];
    float* lines;               /* n*4  cropped-image px, reordered */
    float* normals;             /* n*2  float32(normal) as stored in Segment.msg */
    uint8_t* color;             /* n */
    float* pixels_normalized;   /* n*4 */
    double* ground;             /* n*4 */
    uint8_t* keep;              /* n */
    float* desc;                /* n*72 */
    uint8_t* code;              /* n*32 */
} lfo_frame_out;

/* runs a-1..a-9 on one frame; arrays in out must hold cap segments; returns n (clamped to cap) */
int lfo_process_frame(const lfo_config* c, const uint8_t* bgr_in, lfo_frame_out* out, int cap);

/* host ingest: cv2.imdecode(data, IMREAD_COLOR), ref duckietown_utils/jpg.py:21-31.
 * 0 ok, -6 corrupt, -5 unsupported stream.  bgr: rows x cols x 3 for the size lfo_jpeg_info reports. */
int lfo_jpeg_info(const uint8_t* data, size_t size, int* rows, int* cols, int* ncomp, int* hmax, int* vmax);
int lfo_jpeg_decode(const uint8_t* data, size_t size, uint8_t* bgr);
int lfo_jpeg_coefficients(const uint8_t* data, size_t size, int16_t* qcoef, int cap_blocks, int* n_blocks);

#ifdef __cplusplus
}
#endif
/* anti-instagram colour clustering (kmeans.py:14-47 = scikit-learn's Lloyd iteration from a given init): lf_oracle_kmeans.c */
int lfo_kmeans(const uint8_t* bgr, int n, int k, const double* init, int max_iter, double tol_rel, double* centers,
               int64_t* counts, double* inertia);

#endif
