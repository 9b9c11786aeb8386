// lanefront C ABI (include/lanefront.h): handle, device memory plan, stage sequencing.
// Host-side constants that enter the arithmetic (Gaussian taps, rho, LOG_NT, resize taps,
// HSV division tables, LBD weights) are computed here with the same deterministic
// routines (detmath.h) the kernels use, so they carry the same bits as the CPU oracle's.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <new>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <functional>
#include "common.h"
#include "lsd_bitplane.h"
#include "jpeg_entropy.h"

using namespace lf;

static const char* kStageNames[LF_N_STAGES] = {
    "pre(resize+correct+hsv+masks+dilate)", "canny_nms", "canny_hysteresis", "lsd_blur_resample_grad",
    "lsd_order", "lsd_grow", "segments(normal+project+sanity)", "lbd_gray_blur_sobel", "lbd_descriptor",
    "assoc_pack", "assoc_mfma", "misc", "jpeg(idct+upsample+color)", "lsd_label(components+launch order)" };

struct EvPair { hipEvent_t a, b; int st; };
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

// JPEG ingest state (allocated on first use, grown on demand)
struct JpegState {
    lf::jpeg::WorkerPool pool;                      // persistent host threads
    std::vector<lf::jpeg::FrameCoefs> frames;       // per-frame host coefficient lists (capacity is kept)
    int rows = 0, cols = 0, max_frames = 0;         // geometry the planes were sized for
    DevBuf planes, entries, block_end, hdrs, out;   // device
    DevBuf gh_clean, gh_sub, gh_seg, gh_info, gh_coef;   // entropy decoding on the device (k_jhuff.hip)
    std::vector<int> h_status;
    void* h_stage = nullptr;                        // pinned staging: headers | block_end | entries
    size_t h_stage_bytes = 0;
    hipEvent_t staged = nullptr;                    // the last H2D out of h_stage has completed
    bool staged_pending = false;
    int* h_status_pinned = nullptr;                 // lf_jpeg_decode_batch_gpu_async: the per-frame status, read back behind status_done
    int h_status_pinned_n = 0, status_frames = 0;
    hipEvent_t status_done = nullptr;
};

struct lf_handle {
    lf_config cfg;
    int device = 0;
    int max_frames = 0, cap_lines = 0;
    hipStream_t stream = nullptr;
    char err[512];
    int err_code = 0;
    // geometry
    int Hc = 0, W = 0, Hs = 0, Ws = 0, Ww = 0;
    size_t P = 0, Ps = 0;
    lf_descriptor_params desc_params = { 1, 7, 2, 5 };      // BinaryDescriptor::Params (lf_set_descriptor_params)
    bool lists_lost = false;     // lsd_grow_lists ran out of memory twice: no per-problem lists, run_detect refuses
    int label_items_full = 0;    // LsdParams::label_items of a handle whose lists hold whole images (alloc_lsd_lists lowers it with rec_cap)
    PreParams pre;
    CannyParams canny;
    LsdParams lsd;
    SegParams seg;
    ResizeTables rt;
    int max_nsx = 0, max_nsy = 0;
    // device buffers
    uint8_t *d_frames = nullptr, *d_edges_u8 = nullptr;
    DevBuf dbg_masks;                   // 0/255 byte form of the colour masks, expanded from the bit planes on demand
    size_t frames_bytes = 0;            // allocation behind d_frames
    uint32_t* d_bgr = nullptr;          // corrected working image, BGRX dword per pixel
    uint8_t* d_gray = nullptr;          // BGR2GRAY of it, 1 byte per pixel (read by the LBD gradient stage)
    DevBuf dbg_bgr;
    uint32_t *d_strong = nullptr, *d_weak = nullptr, *d_maskbits = nullptr;
    int *d_sdiv = nullptr, *d_hdiv = nullptr;
    // unordered per-problem records of defined LSD pixels (k_lsd_grad -> k_lsd_order)
    uint32_t* d_raddr = nullptr;
    float* d_rdeg = nullptr;
    double *d_rmod = nullptr, *d_rcs = nullptr, *d_rsn = nullptr;
    float *d_rsd = nullptr, *d_csd = nullptr;      // LsdParams::r_sd, c_sd
    int* d_nrec = nullptr;
    uint8_t* d_zero = nullptr; size_t zero_bytes = 0;   // d_maxgrad | d_nrec | d_nlow | d_tile_count | d_overflow: the counters a batch starts from zero, ONE memset (each memset is a dispatch of its own and waited 0.3 ms in a busy pipeline)
    bool overflow_zeroed = false;
    // lsd_seed_order = OPENCV32 only: pixels with a non-zero but undefined gradient (k_lsd_grad -> k_lsd_seed32)
    uint32_t* d_laddr = nullptr; double* d_lmod = nullptr; int* d_nlow = nullptr;
    unsigned long long *d_sort_a = nullptr, *d_sort_b = nullptr;
    DevBuf dbg_ang, dbg_mod;
    unsigned long long* d_maxgrad = nullptr;
    uint32_t *d_order_a = nullptr, *d_order_b = nullptr, *d_reg = nullptr;
    uint32_t *d_cxy = nullptr, *d_gused = nullptr;
    float* d_cdeg = nullptr;
    double *d_cmod = nullptr, *d_ccs = nullptr, *d_csn = nullptr;
    uint32_t* d_tile_list = nullptr;
    int* d_tile_count = nullptr;
    int* d_row_start = nullptr;
    int *d_norder = nullptr, *d_counts = nullptr, *d_seg_offset = nullptr, *d_frame_offset = nullptr, *d_overflow = nullptr;
    float* d_slot_lines = nullptr;
    uint16_t *d_clabel = nullptr, *d_comp_list = nullptr;     // connected components of the LSD problems (k_lsd_label)
    int* d_comp_count = nullptr;
    int* d_perm = nullptr;
    int* d_comp_key = nullptr;
    float* d_tmp_lines = nullptr;                             // lines in completion order + their seed positions (k_lsd_grow)
    int* d_tmp_tags = nullptr;
    double* d_pend_rec = nullptr;     // pending regions of every LSD problem (k_lsd_grow -> k_lsd_eval): 12 doubles each,
    int* d_pend_tag = nullptr;        // their seed positions,
    int* d_pend_count = nullptr;      // and how many per problem
    int* d_seg_frame = nullptr;
    uint32_t* d_dxy = nullptr;          // LBD gradients, dx | dy << 16 per pixel
    DevBuf dbg_dx, dbg_dy;
    float *d_gauss_g = nullptr, *d_gauss_l = nullptr;
    int *d_xofs = nullptr, *d_y0 = nullptr, *d_y1 = nullptr;
    float *d_xa = nullptr, *d_yb = nullptr;
    // output staging (device side of host-output calls, and the plugin path)
    lf_segments d_out;
    double* d_normals64 = nullptr;
    float* d_centers = nullptr;
    int out_capacity = 0;
    // associator scratch (grown on demand)
    DevBuf a_q, a_m, a_mx, a_mcx, a_best, a_idx, a_dist, a_qn, a_mn;
    DevBuf km_pts, km_lab, km_f64, km_cnt;
    DevBuf kn_hist, kn_count, kn_off, kn_total;       // radiusMatch scratch
    AssocScratch a_ws;
    struct MatcherState* matcher = nullptr;     // BinaryDescriptorMatcher's dataset (lanefront_matcher.inc)
    // pinned host scalars
    int* h_pinned = nullptr;     // [0] total segments, [1] overflow ... [6] entries the per-problem lists would have needed (d_overflow[5])
    const uint8_t* pend_in = nullptr; int pend_n = 0; lf_segments pend_out; bool pend_describe = false;   // the batch in flight (lsd_records_retry)
    int lists_grown = 0;         // times the per-problem lists were reallocated
    int last_frames = 0;
    bool plugin_ready = false;
    bool pending = false;
    // plugin path: what lf_detect_lines hands out is fetched ONCE per image, behind the kernels of lf_set_image, into pinned host
    // memory (the first kPlugEager segments of the SegmentList + the three mask images): lf_detect_lines is then a host copy
    uint8_t* plug_host = nullptr; size_t plug_host_bytes = 0; uint8_t* plug_in = nullptr; size_t plug_in_bytes = 0;
    int plug_eager = 0;
    bool pending_keylines = false;        // the batch in flight is lf_keylines_batch_async's: lf_wait reads the KeyLine state
    bool grow_mixed = false;     // the last batch had problems beyond the slice in numbers (> 1 %): one launch with both kinds of problem code
    int grow_lds_level = 0;      // index into kGrowLdsKb: k_lsd_grow's LDS slice, moved by the share of problems that overflowed it in the last batch
    int detector = LF_DETECTOR_LSD;       // what lf_process_batch runs for a-2 .. a-4 (lf_set_detector)
    lf_edlines_params ed_params;
    int detector_failures = 0;            // frames of the last completed batch on which the EDLines detector gave up
    int tie_rule = LF_TIE_MIHASHER;   // lf_associate: the reference's rule unless lf_set_tie_rule says otherwise
    int env_lds_level = -1;      // LF_GROW_LDS_LEVEL / LF_GROW_MIXED: test and tuning overrides, read when the handle is created, clamped
    int env_mixed = -1;
    int env_bitmap = 1;          // LF_GROW_BITMAP=0: the row-list form of k_lsd_grow (rounds 1 - 3) instead of the bit-plane form (A/B measurements); > 1: see launch_lsd_grow
    int env_kl_lds_lines = 0;    // LF_KL_LDS_LINES (test hook of the KeyLine grouping, lanefront_keylines.inc)
    int pending_problems = 0;
    int pending_capacity = 0;
    std::vector<int> h_counts, h_seg_offset;
    JpegState* jpeg = nullptr;
    struct KlState* kl = nullptr;       // EDLines / KeyLines state (lanefront_keylines.inc), allocated on first use
    struct LsdKlState* lsdkl = nullptr; // LSDDetectorC over octaves (lanefront_lsdkl.inc): sub-handles per pyramid level
    DevBuf m_fo, m_color, m_pn, m_nm, m_gr, m_keep, m_counts, m_boff, m_body, m_bad;   // SegmentList glue scratch
    // profiling
    bool profiling = false;
    std::vector<EvPair> ev_free, ev_used;
    double ms[LF_N_STAGES];
    int32_t launches[LF_N_STAGES];
};

extern "C" void lf_set_error(lf_handle* h, int code, const char* fmt, ...)
{
    if (!h) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(h->err, sizeof(h->err), fmt, ap);
    va_end(ap);
    h->err_code = code;
}

static char g_create_err[512] = "no error";

// LF_ALLOC_TRACE=1: one line per device allocation of a handle on stderr (what the footprint figures in DESIGN.md §3 are made of)
static bool alloc_trace() { static const bool on = [] { const char* e = getenv("LF_ALLOC_TRACE"); return e && *e && *e != '0'; }(); return on; }
template <typename T>
static int dalloc_(lf_handle* h, T** p, size_t count, const char* what)
{
    const size_t bytes = count ? count * sizeof(T) : sizeof(T);
    LF_HIP_CHECK(h, hipMalloc((void**)p, bytes));
    if (alloc_trace()) fprintf(stderr, "lanefront alloc %-28s %12zu B\n", what, bytes);
    return LF_OK;
}
#define dalloc(h, p, count) dalloc_(h, p, count, #p)

static int ensure(lf_handle* h, DevBuf& b, size_t bytes)
{
    if (b.bytes >= bytes) return LF_OK;
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr; b.bytes = 0;
    size_t want = bytes + bytes / 4 + 256;
    LF_HIP_CHECK(h, hipMalloc(&b.p, want));
    b.bytes = want;
    return LF_OK;
}

// Per-stage timing with HIP events recorded on the handle's stream.  Events are only
// recorded here (no host synchronisation inside the pipeline); lf_get_timing resolves them.
static void timing_resolve(lf_handle* h)
{
    for (EvPair& e : h->ev_used) {
        (void)hipEventSynchronize(e.b);
        float t = 0;
        if (hipEventElapsedTime(&t, e.a, e.b) == hipSuccess) h->ms[e.st] += t;
        h->ev_free.push_back(e);
    }
    h->ev_used.clear();
}

struct StageTimer {
    lf_handle* h; int st; EvPair e; bool on;
    StageTimer(lf_handle* h_, int st_) : h(h_), st(st_), on(h_->profiling)
    {
        if (!on) return;
        if (h->ev_free.empty()) {
            if (h->ev_used.size() >= 8192) timing_resolve(h);
            else {
                EvPair n; n.st = 0;
                if (hipEventCreate(&n.a) != hipSuccess || hipEventCreate(&n.b) != hipSuccess) { on = false; return; }
                h->ev_free.push_back(n);
            }
        }
        e = h->ev_free.back(); h->ev_free.pop_back();
        e.st = st;
        (void)hipEventRecord(e.a, h->stream);
    }
    ~StageTimer()
    {
        if (on) { (void)hipEventRecord(e.b, h->stream); h->ev_used.push_back(e); }
        h->launches[st] += 1;
    }
};

static int cv_round_host(double v) { return dm::round_half_even(v); }

static thread_local bool g_lsd_only_create = false;      // set around the lf_create of an LSD-only sub-handle (lanefront_lsdkl.inc)

static int build_params(lf_handle* h)
{
    const lf_config& c = h->cfg;
    h->Hc = c.img_rows - c.top_cutoff;
    h->W = c.img_cols;
    if (h->Hc <= 0 || h->W <= 0 || c.in_rows <= 0 || c.in_cols <= 0 || c.top_cutoff < 0) {
        lf_set_error(h, LF_ERR_BAD_ARG, "bad geometry: in %dx%d img %dx%d cutoff %d", c.in_rows, c.in_cols, c.img_rows,
                     c.img_cols, c.top_cutoff);
        return LF_ERR_BAD_ARG;
    }
    // (the bit planes of the front end are whole 32-bit words per row; a sub-handle that only runs the LSD stages on gray pyramid
    // levels -- lanefront_lsdkl.inc -- never touches them)
    if (h->W % 32 != 0 && !g_lsd_only_create) { lf_set_error(h, LF_ERR_UNSUPPORTED, "img_cols must be a multiple of 32 (got %d)", h->W); return LF_ERR_UNSUPPORTED; }
    h->P = (size_t)h->Hc * h->W;
    h->Ww = (h->W + 31) / 32;
    // ---- pre
    PreParams& p = h->pre;
    memset(&p, 0, sizeof(p));
    p.in_rows = c.in_rows; p.in_cols = c.in_cols; p.img_rows = c.img_rows; p.img_cols = c.img_cols;
    p.top_cutoff = c.top_cutoff; p.Hc = h->Hc; p.W = h->W;
    p.resize = (c.img_rows != c.in_rows) || (c.img_cols != c.in_cols);
    const double fx = (double)c.img_cols / (double)c.in_cols, fy = (double)c.img_rows / (double)c.in_rows;
    p.ifx = 1.0 / fx; p.ify = 1.0 / fy;
    p.identity_ai = 1;
    for (int i = 0; i < 3; ++i) { p.ai_scale[i] = c.ai_scale[i]; p.ai_shift[i] = c.ai_shift[i]; if (c.ai_scale[i] != 1.f || c.ai_shift[i] != 0.f) p.identity_ai = 0; }
    for (int k = 0; k < 4; ++k) for (int ch = 0; ch < 3; ++ch) { p.lo[k][ch] = c.hsv_lo[k][ch]; p.hi[k][ch] = c.hsv_hi[k][ch]; }
    p.ksize = c.dilation_kernel_size;
    if (p.ksize < 1 || p.ksize > kMaxKsize) { lf_set_error(h, LF_ERR_UNSUPPORTED, "dilation_kernel_size %d not in [1,%d]", p.ksize, kMaxKsize); return LF_ERR_UNSUPPORTED; }
    p.r = p.ksize / 2;
    {
        // cv::getStructuringElement(MORPH_ELLIPSE)
        int r = p.ksize / 2, cc = p.ksize / 2;
        double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
        for (int i = 0; i < p.ksize; ++i) {
            int dy = i - r;
            p.j1[i] = 0; p.j2[i] = 0;
            if (abs(dy) <= r) {
                int dx = cv_round_host(cc * sqrt((r * r - dy * dy) * inv_r2));
                p.j1[i] = cc - dx > 0 ? cc - dx : 0;
                p.j2[i] = cc + dx + 1 < p.ksize ? cc + dx + 1 : p.ksize;
            }
        }
        if (p.ksize % 2 == 0) { lf_set_error(h, LF_ERR_UNSUPPORTED, "even dilation kernels are not supported"); return LF_ERR_UNSUPPORTED; }
    }
    // ---- canny
    h->canny.Hc = h->Hc; h->canny.W = h->W; h->canny.Ww = h->Ww;
    double lo = c.canny_lo, hi = c.canny_hi;
    if (lo > hi) { double t = lo; lo = hi; hi = t; }
    h->canny.low = dm::ifloor(lo); h->canny.high = dm::ifloor(hi);
    // ---- LSD
    LsdParams& L = h->lsd;
    memset(&L, 0, sizeof(L));
    L.Hc = h->Hc; L.W = h->W; L.Ww = h->Ww;
    L.scaled = c.lsd_scale != 1.0;
    L.scale = c.lsd_scale;
    if (c.lsd_scale <= 0 || c.lsd_scale > 1.0) { lf_set_error(h, LF_ERR_UNSUPPORTED, "lsd_scale must be in (0,1]"); return LF_ERR_UNSUPPORTED; }
    if (L.scaled) { L.Ws = cv_round_host(h->W * c.lsd_scale); L.Hs = cv_round_host(h->Hc * c.lsd_scale); }
    else { L.Ws = h->W; L.Hs = h->Hc; }
    h->Hs = L.Hs; h->Ws = L.Ws; h->Ps = (size_t)L.Hs * L.Ws;
    if (h->Ps >= (1u << 20) || L.Ws > 65535 || L.Hs > 65535) { lf_set_error(h, LF_ERR_UNSUPPORTED, "scaled LSD image too large"); return LF_ERR_UNSUPPORTED; }
    // LDS limits of the two LDS-resident stages, checked once here so that an unsupported geometry fails at
    // lf_create instead of as a launch error later: region growing keeps the row-start table of its problem in
    // LDS (k_lsd_grow.hip), canny hysteresis sweeps strips of >= 1 row + 2 halo rows (k_canny.hip)
    if ((size_t)((L.Hs + 2) & ~1) * 4 + 512 * 4 + 1024 > 64 * 1024) { lf_set_error(h, LF_ERR_UNSUPPORTED, "scaled LSD image has %d rows: the row table exceeds the LDS of one problem", L.Hs); return LF_ERR_UNSUPPORTED; }
    if ((size_t)h->Hc * h->Ww > 8 * 1024 && (8192 / h->Ww < 1 || (int)((60 * 1024 / 4) / (2 * (size_t)h->Ww)) - 1 < 1)) { lf_set_error(h, LF_ERR_UNSUPPORTED, "img_cols %d: one row of the edge bit planes exceeds the hysteresis strip budget", h->W); return LF_ERR_UNSUPPORTED; }
    if (c.lsd_n_bins < 2 || c.lsd_n_bins > 4096) { lf_set_error(h, LF_ERR_UNSUPPORTED, "lsd_n_bins must be in [2,4096] (a seed is its bin above a 20-bit pixel index in one word)"); return LF_ERR_UNSUPPORTED; }
    if (L.scaled) {
        const double sigma = (c.lsd_scale < 1) ? (c.lsd_sigma_scale / c.lsd_scale) : c.lsd_sigma_scale;
        const double sprec = 3;
        const unsigned hh = (unsigned)ceil(sigma * sqrt(2 * sprec * dm::dlog(10.0)));
        const int n = 1 + 2 * (int)hh;
        if (n > kMaxGaussTaps) { lf_set_error(h, LF_ERR_UNSUPPORTED, "LSD Gaussian needs %d taps (max %d)", n, kMaxGaussTaps); return LF_ERR_UNSUPPORTED; }
        const double scale2X = -0.5 / (sigma * sigma);
        double sum = 0;
        for (int i = 0; i < n; ++i) { double x = i - (n - 1) * 0.5; double t = dm::dexp(scale2X * x * x); L.k[i] = t; sum += t; }
        sum = 1.0 / sum;
        for (int i = 0; i < n; ++i) L.k[i] *= sum;
        L.ntaps = n; L.half = n / 2;
    } else { L.ntaps = 1; L.half = 0; L.k[0] = 1.0; }
    L.prec = 3.14159265358979323846 * c.lsd_ang_th / 180;
    L.p = c.lsd_ang_th / 180;
    L.rho = c.lsd_quant / dm::dsin(L.prec);
    L.log_nt = 5 * (dm::dlog10((double)L.Ws) + dm::dlog10((double)L.Hs)) / 2 + dm::dlog10(11.0);
    L.min_reg_size = (int)(-L.log_nt / dm::dlog10(L.p));
    L.log_eps = c.lsd_log_eps; L.density_th = c.lsd_density_th;
    L.n_bins = c.lsd_n_bins; L.refine = c.lsd_refine; L.cap_lines = h->cap_lines;
    if (c.lsd_seed_order != LF_LSD_SEED_OPENCV30 && c.lsd_seed_order != LF_LSD_SEED_OPENCV32) { lf_set_error(h, LF_ERR_BAD_ARG, "lsd_seed_order %d: LF_LSD_SEED_OPENCV30 or LF_LSD_SEED_OPENCV32", c.lsd_seed_order); return LF_ERR_BAD_ARG; }
    if (c.lsd_seed_order == LF_LSD_SEED_OPENCV32 && !lsd_seed32_supported(L)) { lf_set_error(h, LF_ERR_UNSUPPORTED, "lsd_seed_order OPENCV32: the %dx%d LSD image exceeds the row tables of the sort emulation, or (n_bins - 1) * quant / sin(ang_th) < 361 (k_lsd_seed32.hip)", L.Ws, L.Hs); return LF_ERR_UNSUPPORTED; }
    // component labelling (k_lsd_label): problems of up to label_lds = 6144 defined pixels in LDS (24 KB per workgroup: what one
    // workgroup of k_lsd_grow gives back when it leaves a CU), the larger ones in the problem's region scratch, up to label_items
    // -- a third of the scaled image (every growing wave has a region list of that size in the scratch) and below 2^16 (the
    // labels are u16).  Rounds 2 - 3 had every problem's tables in LDS, 48 KB per workgroup growing with the workload to 144 KB.
    L.label_lds = kLabelLds;
    {
        // images whose bit plane is larger than that anyway (1080p: 124 KB, one workgroup per CU): the LDS form for every problem
        // that fits the same request
        const size_t plane = bitplane_lds_words(h->Ps) * 4;
        if (plane <= 150 * 1024 && plane / 4 > (size_t)L.label_lds) L.label_lds = (int)((plane / 4 < 65534 ? plane / 4 : 65534) & ~(size_t)1);
    }
    L.label_items = (int)(h->Ps / 3 / 1024) * 1024;
    if (L.label_items > 64512) L.label_items = 64512;
    if (L.label_items < kLabelItems) L.label_items = kLabelItems;
    L.label_items_max = L.label_items;
    h->label_items_full = L.label_items;
    L.rec_cap = (int)h->Ps;                               // (alloc_buffers chooses the batch handles' starting capacity)
    // ---- segments
    SegParams& S = h->seg;
    memset(&S, 0, sizeof(S));
    S.Hc = h->Hc; S.W = h->W; S.img_rows = c.img_rows; S.img_cols = c.img_cols; S.top_cutoff = c.top_cutoff;
    S.cap_lines = h->cap_lines;
    S.rx = 1.0 / (double)c.img_cols; S.ry = 1.0 / (double)c.img_rows; S.cut = (double)c.top_cutoff;
    S.cw = (double)c.cam_w; S.ch = (double)c.cam_h;
    memcpy(S.H, c.H, sizeof(S.H)); memcpy(S.K, c.K, sizeof(S.K)); memcpy(S.D, c.D, sizeof(S.D));
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int t = 0; t < 3; ++t) s += c.P[4 * i + t] * c.R[3 * t + j];
            S.RR[3 * i + j] = s;
        }
    S.lanewidth = c.lanewidth; S.linewidth_white = c.linewidth_white; S.linewidth_yellow = c.linewidth_yellow;
    S.d_min = c.d_min; S.d_max = c.d_max; S.phi_min = c.phi_min; S.phi_max = c.phi_max;
    return LF_OK;
}

// LBD Gaussian weights for a band width w (binary_descriptor_custom.cpp:217-259 = setWidthOfBand :134-176; the integer divisions in u and
// sigma are the reference's): 9 w global weights F_g, 3 w local weights F_l, into the handle's device tables
static int lbd_weights(lf_handle* h, int w)
{
    std::vector<float> gg(9 * (size_t)w), gl(3 * (size_t)w);
    double u = (w * 3 - 1) / 2;
    double sigma = (w * 2 + 1) / 2;
    double inv = -1 / (2 * sigma * sigma);
    for (int i = 0; i < 3 * w; ++i) { double d = i - u; gl[i] = (float)dm::dexp(d * d * inv); }
    u = (9 * w - 1) / 2;
    sigma = u;
    inv = -1 / (2 * sigma * sigma);
    for (int i = 0; i < 9 * w; ++i) { double d = i - u; gg[i] = (float)dm::dexp(d * d * inv); }
    LF_HIP_CHECK(h, hipMemcpy(h->d_gauss_g, gg.data(), gg.size() * sizeof(float), hipMemcpyHostToDevice));
    LF_HIP_CHECK(h, hipMemcpy(h->d_gauss_l, gl.data(), gl.size() * sizeof(float), hipMemcpyHostToDevice));
    return LF_OK;
}

static int upload_tables(lf_handle* h)
{
    // HSV fixed-point division tables (OpenCV RGB2HSV_b)
    std::vector<int> sdiv(256), hdiv(256);
    sdiv[0] = hdiv[0] = 0;
    for (int i = 1; i < 256; ++i) {
        sdiv[i] = cv_round_host((255 << 12) / (1.0 * i));
        hdiv[i] = cv_round_host((180 << 12) / (6.0 * i));
    }
    if (dalloc(h, &h->d_sdiv, 256) || dalloc(h, &h->d_hdiv, 256)) return LF_ERR_HIP;
    LF_HIP_CHECK(h, hipMemcpy(h->d_sdiv, sdiv.data(), 256 * sizeof(int), hipMemcpyHostToDevice));
    LF_HIP_CHECK(h, hipMemcpy(h->d_hdiv, hdiv.data(), 256 * sizeof(int), hipMemcpyHostToDevice));
    // LSD resize tables (cv::resize INTER_LINEAR, CV_64F)
    const LsdParams& L = h->lsd;
    const int Ws = L.Ws, Hs = L.Hs, W = h->W, Hc = h->Hc;
    std::vector<int> xofs(Ws), y0(Hs), y1(Hs);
    std::vector<float> xa(2 * (size_t)Ws), yb(2 * (size_t)Hs);
    int xmax = Ws;
    const double scale_x = 1.0 / L.scale, scale_y = 1.0 / L.scale;
    for (int dx = 0; dx < Ws; ++dx) {
        float fx; int sx;
        if (L.scaled) {
            fx = (float)((dx + 0.5) * scale_x - 0.5);
            sx = dm::ifloor((double)fx);
            fx -= sx;
        } else { fx = 0.f; sx = dx; }
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= W) {
            if (dx < xmax) xmax = dx;
            if (sx >= W - 1) { fx = 0; sx = W - 1; }
        }
        xofs[dx] = sx; xa[2 * dx] = 1.f - fx; xa[2 * dx + 1] = fx;
    }
    for (int dy = 0; dy < Hs; ++dy) {
        float fy; int sy;
        if (L.scaled) {
            fy = (float)((dy + 0.5) * scale_y - 0.5);
            sy = dm::ifloor((double)fy);
            fy -= sy;
        } else { fy = 0.f; sy = dy; }
        yb[2 * dy] = 1.f - fy; yb[2 * dy + 1] = fy;
        y0[dy] = sy < 0 ? 0 : (sy > Hc - 1 ? Hc - 1 : sy);
        y1[dy] = sy + 1 < 0 ? 0 : (sy + 1 > Hc - 1 ? Hc - 1 : sy + 1);
    }
    // LDS footprint of the worst tile
    const int GT = 32;
    int mx = 0, my = 0;
    for (int X0 = 0; X0 < Ws; X0 += GT) {
        int X1 = X0 + GT < Ws - 1 ? X0 + GT : Ws - 1;
        int lo = xofs[X0], hi = xofs[X1] + 1 < W - 1 ? xofs[X1] + 1 : W - 1;
        if (hi - lo + 1 > mx) mx = hi - lo + 1;
    }
    for (int Y0 = 0; Y0 < Hs; Y0 += GT) {
        int Y1 = Y0 + GT < Hs - 1 ? Y0 + GT : Hs - 1;
        int lo = y0[Y0], hi = y1[Y1];
        if (hi - lo + 1 > my) my = hi - lo + 1;
    }
    h->max_nsx = mx; h->max_nsy = my;
    {
        const int hh = L.half;
        // same carve as launch_lsd_grad: F|Hb|pixel list share one region, Bl|Sc the other
        const size_t szF = (size_t)(my + 2 * hh) * mx, szBl = (size_t)my * mx;
        const size_t szHb = (size_t)my * (GT + 1), szSc = (size_t)(GT + 1) * (GT + 1);
        size_t regA = szF > szHb ? szF : szHb;
        if (regA < (size_t)2 * GT * GT) regA = (size_t)2 * GT * GT;
        const size_t regB = szBl > szSc ? szBl : szSc;
        size_t lds = sizeof(double) * (regA + regB);
        if (lds > 64 * 1024) { lf_set_error(h, LF_ERR_UNSUPPORTED, "lsd_scale %.3f needs %zu B of LDS per tile (max 65536)", L.scale, lds); return LF_ERR_UNSUPPORTED; }
    }
    if (dalloc(h, &h->d_xofs, Ws) || dalloc(h, &h->d_y0, Hs) || dalloc(h, &h->d_y1, Hs) || dalloc(h, &h->d_xa, 2 * (size_t)Ws) ||
        dalloc(h, &h->d_yb, 2 * (size_t)Hs)) return LF_ERR_HIP;
    LF_HIP_CHECK(h, hipMemcpy(h->d_xofs, xofs.data(), Ws * sizeof(int), hipMemcpyHostToDevice));
    LF_HIP_CHECK(h, hipMemcpy(h->d_y0, y0.data(), Hs * sizeof(int), hipMemcpyHostToDevice));
    LF_HIP_CHECK(h, hipMemcpy(h->d_y1, y1.data(), Hs * sizeof(int), hipMemcpyHostToDevice));
    LF_HIP_CHECK(h, hipMemcpy(h->d_xa, xa.data(), 2 * (size_t)Ws * sizeof(float), hipMemcpyHostToDevice));
    LF_HIP_CHECK(h, hipMemcpy(h->d_yb, yb.data(), 2 * (size_t)Hs * sizeof(float), hipMemcpyHostToDevice));
    h->rt.xofs = h->d_xofs; h->rt.xa = h->d_xa; h->rt.y0 = h->d_y0; h->rt.y1 = h->d_y1; h->rt.yb = h->d_yb; h->rt.xmax = xmax;
    const int mw = lbd_max_width_of_band();
    if (dalloc(h, &h->d_gauss_g, 9 * (size_t)mw) || dalloc(h, &h->d_gauss_l, 3 * (size_t)mw)) return LF_ERR_HIP;
    return lbd_weights(h, h->desc_params.width_of_band);
}

// BinaryDescriptor::Params on a handle (binary_descriptor_custom.cpp:108-200)
extern "C" void lf_descriptor_default_params(lf_descriptor_params* p)
{
    if (!p) return;
    p->num_of_octave = 1; p->width_of_band = 7; p->reduction_ratio = 2; p->ksize = 5;      // :110-116
}

extern "C" int lf_get_descriptor_params(lf_handle* h, lf_descriptor_params* p)
{
    if (!h) return LF_ERR_NOT_INITIALISED;
    if (!p) { lf_set_error(h, LF_ERR_BAD_ARG, "lf_get_descriptor_params: null argument"); return LF_ERR_BAD_ARG; }
    *p = h->desc_params;
    return LF_OK;
}

extern "C" int lf_set_descriptor_params(lf_handle* h, const lf_descriptor_params* p)
{
    if (!h) return LF_ERR_NOT_INITIALISED;
    if (!p) { lf_set_error(h, LF_ERR_BAD_ARG, "lf_set_descriptor_params: null argument"); return LF_ERR_BAD_ARG; }
    if (h->pending) { lf_set_error(h, LF_ERR_BAD_ARG, "a batch is in flight on this handle: call lf_wait first"); return LF_ERR_BAD_ARG; }
    if (p->width_of_band < 1 || p->width_of_band > lbd_max_width_of_band()) {
        lf_set_error(h, LF_ERR_UNSUPPORTED, "widthOfBand %d outside 1..%d", p->width_of_band, lbd_max_width_of_band());
        return LF_ERR_UNSUPPORTED;
    }
    if (p->ksize < 1 || p->ksize > 31 || !(p->ksize & 1)) { lf_set_error(h, LF_ERR_BAD_ARG, "ksize %d: an odd size in 1..31 (cv::GaussianBlur asserts the oddness)", p->ksize); return LF_ERR_BAD_ARG; }
    if (p->num_of_octave < 1 || p->num_of_octave > LF_MAX_OCTAVES || p->reduction_ratio < 1) {
        lf_set_error(h, LF_ERR_BAD_ARG, "lf_set_descriptor_params: numOfOctave_ outside 1..%d or reductionRatio < 1", LF_MAX_OCTAVES);
        return LF_ERR_BAD_ARG;
    }
    LF_HIP_CHECK(h, hipSetDevice(h->device));
    LF_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    if (p->width_of_band != h->desc_params.width_of_band) { const int rc = lbd_weights(h, p->width_of_band); if (rc != LF_OK) return rc; }
    h->desc_params = *p;
    return LF_OK;
}

// the arrays whose stride is LsdParams::rec_cap: records, sort scratch, seed lists, compact arrays, labels, the region scratch, low records
static void free_lsd_lists(lf_handle* h)
{
    void* ptrs[] = { h->d_raddr, h->d_rdeg, h->d_rmod, h->d_rcs, h->d_rsn, h->d_sort_a, h->d_sort_b, h->d_order_a, h->d_order_b, h->d_cxy, h->d_cdeg,
                     h->d_cmod, h->d_ccs, h->d_reg, h->d_clabel, h->d_laddr, h->d_lmod, h->d_rsd, h->d_csd };
    for (void* p : ptrs) if (p) (void)hipFree(p);
    h->d_rsd = nullptr; h->d_csd = nullptr; h->lsd.r_sd = nullptr; h->lsd.c_sd = nullptr;
    h->d_raddr = nullptr; h->d_rdeg = nullptr; h->d_rmod = nullptr; h->d_rcs = nullptr; h->d_rsn = nullptr; h->d_sort_a = nullptr; h->d_sort_b = nullptr;
    h->d_order_a = nullptr; h->d_order_b = nullptr; h->d_cxy = nullptr; h->d_cdeg = nullptr; h->d_cmod = nullptr; h->d_ccs = nullptr; h->d_csn = nullptr;
    h->d_reg = nullptr; h->d_clabel = nullptr; h->d_laddr = nullptr; h->d_lmod = nullptr;
}

static int alloc_lsd_lists(lf_handle* h, int rec_cap)
{
    const size_t nprob = (size_t)h->max_frames * 3, S = (size_t)rec_cap;
    LsdParams& L = h->lsd;
    L.rec_cap = rec_cap;
    // (components are only kept apart for problems of up to label_items defined pixels; no problem has more than rec_cap)
    // (... but not below kLabelItems: the labelling kernel's LDS form and the growing waves' scratch slices are laid out for that)
    const int li = rec_cap > kLabelItems ? rec_cap : kLabelItems;
    L.label_items = h->label_items_full < li ? h->label_items_full : li;
    L.label_items_max = L.label_items;
    if (dalloc(h, &h->d_raddr, nprob * S) || dalloc(h, &h->d_rdeg, nprob * S) || dalloc(h, &h->d_rmod, nprob * S) ||
        dalloc(h, &h->d_rcs, nprob * S) || dalloc(h, &h->d_rsn, nprob * S) || dalloc(h, &h->d_sort_a, nprob * S) || dalloc(h, &h->d_sort_b, nprob * S) ||
        dalloc(h, &h->d_order_a, nprob * S) || dalloc(h, &h->d_order_b, nprob * S) || dalloc(h, &h->d_cxy, nprob * S) || dalloc(h, &h->d_cdeg, nprob * S) ||
        dalloc(h, &h->d_cmod, nprob * S) || dalloc(h, &h->d_ccs, nprob * S * 2) || dalloc(h, &h->d_reg, nprob * lsd_grow_reg_stride(L)) ||
        dalloc(h, &h->d_clabel, nprob * S))
        return LF_ERR_HIP;
    h->d_csn = h->d_ccs + 1;          // (cos, sin) pairs in one array: k_lsd_order.hip
    if (dalloc(h, &h->d_rsd, nprob * S * 2) || dalloc(h, &h->d_csd, nprob * S * 2)) return LF_ERR_HIP;
    L.r_sd = h->d_rsd; L.c_sd = h->d_csd;
    if (h->cfg.lsd_seed_order == LF_LSD_SEED_OPENCV32 && (dalloc(h, &h->d_laddr, nprob * S) || dalloc(h, &h->d_lmod, nprob * S)))
        return LF_ERR_HIP;
    return LF_OK;
}

static int alloc_buffers(lf_handle* h)
{
    const size_t B = (size_t)h->max_frames, P = h->P, Ps = h->Ps;
    const size_t in_px = (size_t)h->cfg.in_rows * h->cfg.in_cols;
    const size_t nprob = B * 3;
    const size_t cap = nprob * (size_t)h->cap_lines;
    // the plugin path (lf_set_image) stages one WORKING image here, which is larger than an input frame when
    // img_size > in_size
    h->frames_bytes = B * in_px * 3 > P * 3 ? B * in_px * 3 : P * 3;
    if (dalloc(h, &h->d_frames, h->frames_bytes) || dalloc(h, &h->d_bgr, B * P) || dalloc(h, &h->d_gray, B * P) || 
        dalloc(h, &h->d_edges_u8, B * P) || dalloc(h, &h->d_strong, B * h->Hc * h->Ww) || dalloc(h, &h->d_weak, B * h->Hc * h->Ww) || dalloc(h, &h->d_maskbits, nprob * h->Hc * h->Ww) ||
        dalloc(h, &h->d_tile_list, nprob * (size_t)(((h->Ws + 31) / 32) * ((h->Hs + 31) / 32))) ||
        dalloc(h, &h->d_gused, nprob * ((Ps + 31) / 32)) || dalloc(h, &h->d_row_start, nprob * (size_t)(h->Hs + 1)) ||
        dalloc(h, &h->d_comp_list, nprob * (size_t)kCompCap) || dalloc(h, &h->d_comp_count, nprob) || dalloc(h, &h->d_perm, nprob) || dalloc(h, &h->d_comp_key, nprob) ||
        dalloc(h, &h->d_tmp_lines, cap * 4) || dalloc(h, &h->d_tmp_tags, cap) ||
        dalloc(h, &h->d_pend_rec, nprob * (size_t)lsd_grow_pend_cap(h->lsd) * 12 + 2) || dalloc(h, &h->d_pend_tag, nprob * (size_t)lsd_grow_pend_cap(h->lsd) + 1) || dalloc(h, &h->d_pend_count, nprob) || dalloc(h, &h->d_norder, nprob) ||
        dalloc(h, &h->d_counts, nprob) || dalloc(h, &h->d_seg_offset, nprob + 1) || dalloc(h, &h->d_frame_offset, B + 1) ||
        dalloc(h, &h->d_slot_lines, cap * 4) || dalloc(h, &h->d_seg_frame, cap) ||
        dalloc(h, &h->d_dxy, B * P) || dalloc(h, &h->d_normals64, cap * 2) ||
        dalloc(h, &h->d_centers, cap * 2))
        return LF_ERR_HIP;
    // The per-problem lists.  A batch handle starts with an eighth of the LSD image per problem (a lane frame's colour has 3 - 6 % of its
    // pixels defined, a camera frame's 10 - 20 %) and grows when a batch needs more (lsd_records_retry); handles of a few frames and the
    // LSD-only sub-handles (gray images: every pixel can be defined) hold whole images.  LF_LSD_RECORDS=<entries> | full overrides.
    {
        size_t cap = Ps;
        if (h->max_frames > 16 && !g_lsd_only_create) cap = (Ps / 8 + 4095) / 4096 * 4096;
        const char* e = g_lsd_only_create ? nullptr : getenv("LF_LSD_RECORDS");
        if (e && *e) { if (!strcmp(e, "full")) cap = Ps; else if (atol(e) > 0) cap = (size_t)atol(e); }
        else if (cap < 8192) cap = 8192;
        if (cap < 1024) cap = 1024;
        if (cap > Ps) cap = Ps;
        const int rc = alloc_lsd_lists(h, (int)cap);
        if (rc != LF_OK) return rc;
    }
    {
        h->zero_bytes = nprob * 8 + nprob * 4 + nprob * 4 + 16 + 32;
        if (dalloc(h, &h->d_zero, h->zero_bytes)) return LF_ERR_HIP;
        h->d_maxgrad = reinterpret_cast<unsigned long long*>(h->d_zero);
        h->d_nrec = reinterpret_cast<int*>(h->d_zero + nprob * 8);
        int* nlow = h->d_nrec + nprob;
        if (h->cfg.lsd_seed_order == LF_LSD_SEED_OPENCV32) h->d_nlow = nlow;
        h->d_tile_count = nlow + nprob;
        h->d_overflow = h->d_tile_count + 4;
        LF_HIP_CHECK(h, hipMemset(h->d_zero, 0, h->zero_bytes));
    }
    h->out_capacity = (int)cap;
    lf_segments& o = h->d_out;
    memset(&o, 0, sizeof(o));
    o.capacity = (int)cap;
    if (dalloc(h, &o.lines, cap * 4) || dalloc(h, &o.normals, cap * 2) || dalloc(h, &o.color, cap) ||
        dalloc(h, &o.pixels_normalized, cap * 4) || dalloc(h, &o.ground, cap * 4) || dalloc(h, &o.keep, cap) ||
        dalloc(h, &o.desc, cap * 72) || dalloc(h, &o.code, cap * 32))
        return LF_ERR_HIP;
    o.frame_offset = h->d_frame_offset;
    LF_HIP_CHECK(h, hipHostMalloc((void**)&h->h_pinned, 16 * sizeof(int)));
    return LF_OK;
}

extern "C" int lf_abi_version(void) { return LF_ABI_VERSION; }

extern "C" int lf_get_stream(lf_handle* h, void** hip_stream)
{
    if (!h) return LF_ERR_NOT_INITIALISED;
    if (!hip_stream) { lf_set_error(h, LF_ERR_BAD_ARG, "lf_get_stream: null argument"); return LF_ERR_BAD_ARG; }
    *hip_stream = static_cast<void*>(h->stream);
    return LF_OK;
}

extern "C" const char* lf_last_error(const lf_handle* h) { return h ? h->err : g_create_err; }

extern "C" const char* lf_stage_name(int stage) { return (stage >= 0 && stage < LF_N_STAGES) ? kStageNames[stage] : "?"; }

struct KlState;
static void kl_free(KlState* k);
struct LsdKlState;
static void lsdkl_free(LsdKlState* k);

static void matcher_free(struct MatcherState* m);
extern "C" void lf_destroy(lf_handle* h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void* ptrs[] = { h->d_frames, h->d_bgr, h->d_gray, h->dbg_masks.p, h->d_edges_u8, h->d_strong, h->d_weak, h->d_maskbits, h->d_sdiv, h->d_hdiv,
                     h->d_zero, h->dbg_ang.p, h->dbg_mod.p, h->d_gused, h->d_row_start, h->d_tile_list,
                     h->d_norder, h->d_counts, h->d_seg_offset, h->d_frame_offset, h->d_slot_lines,
                     h->d_seg_frame, h->d_comp_list, h->d_comp_count, h->d_perm, h->d_comp_key, h->d_tmp_lines, h->d_tmp_tags, h->d_pend_rec, h->d_pend_tag, h->d_pend_count, h->d_dxy, h->dbg_dx.p, h->dbg_dy.p, h->d_gauss_g, h->d_gauss_l, h->d_xofs, h->d_y0, h->d_y1,
                     h->d_xa, h->d_yb, h->d_out.lines, h->d_out.normals, h->d_out.color, h->d_out.pixels_normalized,
                     h->d_out.ground, h->d_out.keep, h->d_out.desc, h->d_out.code, h->d_normals64, h->d_centers,
                     h->km_pts.p, h->km_lab.p, h->km_f64.p, h->km_cnt.p, h->kn_hist.p, h->kn_count.p, h->kn_off.p, h->kn_total.p, h->a_q.p, h->a_m.p, h->a_mx.p, h->a_mcx.p, h->a_best.p, h->a_idx.p, h->a_dist.p, h->a_qn.p, h->a_mn.p, h->dbg_bgr.p };
    for (void* p : ptrs) if (p) (void)hipFree(p);
    free_lsd_lists(h);
    assoc_scratch_free(h->a_ws);
    if (h->h_pinned) (void)hipHostFree(h->h_pinned);
    if (h->plug_host) (void)hipHostFree(h->plug_host);
    if (h->plug_in) (void)hipHostFree(h->plug_in);
    for (DevBuf* b : { &h->m_fo, &h->m_color, &h->m_pn, &h->m_nm, &h->m_gr, &h->m_keep, &h->m_counts, &h->m_boff, &h->m_body, &h->m_bad })
        if (b->p) (void)hipFree(b->p);
    if (h->jpeg) {
        JpegState* j = h->jpeg;
        for (DevBuf* b : { &j->planes, &j->entries, &j->block_end, &j->hdrs, &j->out, &j->gh_clean, &j->gh_sub, &j->gh_seg, &j->gh_info, &j->gh_coef }) if (b->p) (void)hipFree(b->p);
        if (j->h_stage) (void)hipHostFree(j->h_stage);
        if (j->staged) (void)hipEventDestroy(j->staged);
        if (j->status_done) (void)hipEventDestroy(j->status_done);
        if (j->h_status_pinned) (void)hipHostFree(j->h_status_pinned);
        delete j;
    }
    kl_free(h->kl);
    lsdkl_free(h->lsdkl);
    matcher_free(h->matcher);
    timing_resolve(h);
    for (EvPair& e : h->ev_free) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int lf_create(const lf_config* cfg, int device_id, int max_frames, int max_lines_per_color, lf_handle** out)
{
    if (!cfg || !out || max_frames < 1 || max_lines_per_color < 1) {
        snprintf(g_create_err, sizeof(g_create_err), "lf_create: bad argument");
        return LF_ERR_BAD_ARG;
    }
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        snprintf(g_create_err, sizeof(g_create_err), "lf_create: no HIP device (%s); lanefront has no CPU fallback",
                 e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return LF_ERR_HIP;
    }
    if (device_id < 0 || device_id >= ndev) {
        snprintf(g_create_err, sizeof(g_create_err), "lf_create: device %d out of range (%d devices)", device_id, ndev);
        return LF_ERR_BAD_ARG;
    }
    lf_handle* h = new (std::nothrow) lf_handle();
    if (!h) return LF_ERR_HIP;
    h->cfg = *cfg; h->device = device_id; h->max_frames = max_frames; h->cap_lines = max_lines_per_color;
    h->err[0] = 0;
    memset(h->ms, 0, sizeof(h->ms)); memset(h->launches, 0, sizeof(h->launches));
    if (const char* ev = getenv("LF_GROW_LDS_LEVEL")) { const int v = atoi(ev); h->env_lds_level = v < 0 ? 0 : (v > 2 ? 2 : v); }
    if (const char* ev = getenv("LF_GROW_MIXED")) h->env_mixed = atoi(ev) != 0 ? 1 : 0;
    if (const char* ev = getenv("LF_GROW_BITMAP")) { const int v = atoi(ev); h->env_bitmap = v < 0 ? 0 : v; }       // > 1: that many USED bits (tests)
    if (const char* ev = getenv("LF_KL_LDS_LINES")) { const int v = atoi(ev); h->env_kl_lds_lines = v < 1 ? 1 : (v > 4096 ? 4096 : v); }
    int rc = LF_OK;
    do {
        if (hipSetDevice(device_id) != hipSuccess) { lf_set_error(h, LF_ERR_HIP, "hipSetDevice(%d) failed", device_id); rc = LF_ERR_HIP; break; }
        if ((rc = build_params(h)) != LF_OK) break;
        if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { lf_set_error(h, LF_ERR_HIP, "hipStreamCreate failed"); rc = LF_ERR_HIP; break; }
        if ((rc = upload_tables(h)) != LF_OK) break;
        if ((rc = alloc_buffers(h)) != LF_OK) break;
    } while (0);
    if (rc != LF_OK) {
        snprintf(g_create_err, sizeof(g_create_err), "%s", h->err);
        lf_destroy(h);
        return rc;
    }
    *out = h;
    return LF_OK;
}

extern "C" int lf_synchronize(lf_handle* h)
{
    if (!h) return LF_ERR_NOT_INITIALISED;
    LF_HIP_CHECK(h, hipSetDevice(h->device));
    LF_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    return LF_OK;
}

// detect stages a-1..a-4 on device-resident frames
static int run_detect(lf_handle* h, const uint8_t* d_frames, int n, bool from_working_image)
{
    hipStream_t s = h->stream;
    if (h->lists_lost) { lf_set_error(h, LF_ERR_HIP, "the handle lost its LSD lists to an out-of-memory growth (lf_wait / lf_set_image reported it)"); return LF_ERR_HIP; }
    h->overflow_zeroed = false;          // (set at the successful END only: an error exit must not leave run_segments believing the overflow words are zero)
    PreParams pp = h->pre;
    if (from_working_image) {
        // plugin path: the caller already resized, cropped and colour-corrected (line_detector_node.py:163-180)
        pp.in_rows = h->Hc; pp.in_cols = h->W; pp.img_rows = h->Hc; pp.img_cols = h->W; pp.top_cutoff = 0;
        pp.resize = 0;
        for (int i = 0; i < 3; ++i) { pp.ai_scale[i] = 1.f; pp.ai_shift[i] = 0.f; }
        pp.identity_ai = 1;
    }
    { StageTimer t(h, ST_PRE); launch_pre(pp, d_frames, n, h->d_bgr, h->d_gray, h->d_maskbits, h->d_sdiv, h->d_hdiv, s); }
    { StageTimer t(h, ST_CANNY); launch_canny(h->canny, h->d_bgr, n, h->d_strong, h->d_weak, s); }
    {
        StageTimer t(h, ST_HYST);
        if (launch_hysteresis(h->canny, n, h->d_strong, h->d_weak, s) != 0) {
            lf_set_error(h, LF_ERR_UNSUPPORTED, "canny hysteresis: a %dx%d working image does not fit the LDS-resident strips", h->Hc, h->W);
            return LF_ERR_UNSUPPORTED;
        }
    }
    {
        StageTimer t(h, ST_LSD_GRAD);
        LF_HIP_CHECK(h, hipMemsetAsync(h->d_zero, 0, h->zero_bytes, s));           // every counter of the batch (see d_zero)
        launch_lsd_grad(h->lsd, h->rt, n, h->d_strong, h->d_maskbits, h->d_raddr, h->d_rdeg, h->d_rmod, h->d_rcs, h->d_rsn, h->d_nrec,
                        h->d_maxgrad, h->max_nsx, h->max_nsy, h->d_tile_list, h->d_tile_count, h->d_laddr, h->d_lmod, h->d_nlow, h->d_overflow + 5, true, s);
    }
    {
        StageTimer t(h, ST_LSD_ORDER);
        launch_lsd_order(h->lsd, n, h->d_raddr, h->d_rdeg, h->d_rmod, h->d_rcs, h->d_rsn, h->d_nrec, h->d_maxgrad, h->d_sort_a, h->d_sort_b, h->d_order_a, h->d_order_b, h->d_norder, h->d_cxy, h->d_cdeg, h->d_cmod, h->d_ccs, h->d_csn, h->d_row_start, s);
        // OpenCV >= 3.2: the seeds in the order std::sort leaves them in (the compact arrays and row starts stay as they are)
        if (h->cfg.lsd_seed_order == LF_LSD_SEED_OPENCV32)
            launch_lsd_seed32(h->lsd, n, h->d_nrec, h->d_norder, h->d_overflow + 5, h->d_maxgrad, h->d_cxy, h->d_cmod, h->d_laddr, h->d_lmod, h->d_nlow, h->d_sort_a, h->d_sort_b, h->d_order_a, h->d_order_b, 0, s);
    }
    {
        StageTimer t(h, ST_LSD_LABEL);          // (its own stage since round 6: two brackets of different content under one name made the average meaningless)
        launch_lsd_label(h->lsd, n, h->d_norder, h->d_cxy, h->d_row_start, h->d_clabel, h->d_comp_list, h->d_comp_count, h->d_comp_key, h->d_reg, s);
        static const bool no_rank = getenv("LF_DIAG_NO_RANK") != nullptr;
        if (!no_rank) launch_lsd_rank(n * 3, h->d_comp_key, h->d_perm, s);
    }
    static const char* diag_skip = getenv("LF_DIAG_SKIP");     // diagnostic only (what-if timing, results are wrong): "grow"
    if (diag_skip && strstr(diag_skip, "grow")) LF_HIP_CHECK(h, hipMemsetAsync(h->d_counts, 0, (size_t)n * 3 * sizeof(int), s));
    else {
        StageTimer t(h, ST_LSD_GROW);
        static const bool no_rank = getenv("LF_DIAG_NO_RANK") != nullptr;
        const int env_lds_level = h->env_lds_level, env_mixed = h->env_mixed;
        launch_lsd_grow(h->lsd, n, h->d_order_a, h->d_norder, h->d_cxy, h->d_cdeg, h->d_cmod, h->d_ccs, h->d_csn, h->d_row_start,
                        h->d_clabel, h->d_comp_list, h->d_comp_count, kCompCap, h->d_reg, h->d_gused, h->d_tmp_lines, h->d_tmp_tags,
                        h->d_slot_lines, h->d_counts, no_rank ? nullptr : h->d_perm, h->d_pend_rec, h->d_pend_tag, h->d_pend_count,
                        kGrowLdsKb[env_lds_level >= 0 ? env_lds_level : h->grow_lds_level],
                        env_mixed >= 0 ? env_mixed != 0 : h->grow_mixed,
                        h->env_bitmap, s);
    }
    LF_HIP_CHECK(h, hipGetLastError());
    h->last_frames = n;
    h->overflow_zeroed = true;           // the batch's one memset (above) covered the overflow words
    return LF_OK;
}

static int run_detect_edlines(lf_handle* h, const uint8_t* d_frames, int n);     // lanefront_keylines.inc
static void keylines_pending_result(lf_handle* h, int* total, int* overflow);

static int run_segments(lf_handle* h, int n, lf_segments dev_out, bool describe)
{
    hipStream_t s = h->stream;
    {
        StageTimer t(h, ST_SEGMENTS);
        if (!h->overflow_zeroed) LF_HIP_CHECK(h, hipMemsetAsync(h->d_overflow, 0, 4 * sizeof(int), s));
        h->overflow_zeroed = false;
        launch_seg_offsets(n, h->cap_lines, h->d_counts, h->d_seg_offset, dev_out.frame_offset ? dev_out.frame_offset : h->d_frame_offset,
                           h->d_overflow, h->d_norder, lsd_grow_def_lds(h->lsd, kGrowLdsKb[0]), lsd_grow_def_lds(h->lsd, kGrowLdsKb[1]), s);
        launch_segments(h->seg, n, h->d_slot_lines, h->d_counts, h->d_seg_offset, h->d_maskbits, h->Ww, dev_out, h->d_seg_frame,
                        h->d_normals64, h->d_centers, s);
    }
    if (describe) {
        { StageTimer t(h, ST_LBD_GRAD); launch_lbd_grad(h->Hc, h->W, n, h->d_gray, h->d_dxy, s); }
        {
            StageTimer t(h, ST_LBD);
            int cap = dev_out.capacity < n * 3 * h->cap_lines ? dev_out.capacity : n * 3 * h->cap_lines;
            launch_lbd(h->Hc, h->W, cap, h->d_seg_offset + n * 3, dev_out.lines, h->d_seg_frame, h->d_dxy,
                       h->d_gauss_g, h->d_gauss_l, dev_out.desc, dev_out.code, s, h->desc_params.width_of_band);
        }
    }
    LF_HIP_CHECK(h, hipGetLastError());
    return LF_OK;
}

// queue a-1..a-9 for a batch on the handle's stream; device outputs only; no host sync
extern "C" int lf_process_batch_async(lf_handle* h, const uint8_t* frames, int n_frames, int frames_on_device,
                                      lf_segments* out_dev, int describe)
{
    if (!h) return LF_ERR_NOT_INITIALISED;
    if (!frames || !out_dev || n_frames < 1) { lf_set_error(h, LF_ERR_BAD_ARG, "lf_process_batch: null argument or n_frames < 1"); return LF_ERR_BAD_ARG; }
    if (n_frames > h->max_frames) { lf_set_error(h, LF_ERR_CAPACITY, "n_frames %d exceeds max_frames %d", n_frames, h->max_frames); return LF_ERR_CAPACITY; }
    if (describe && (!out_dev->lines)) { lf_set_error(h, LF_ERR_BAD_ARG, "describe needs out->lines"); return LF_ERR_BAD_ARG; }
    if (h->pending) { lf_set_error(h, LF_ERR_BAD_ARG, "a batch is already in flight on this handle: call lf_wait first"); return LF_ERR_BAD_ARG; }
    LF_HIP_CHECK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const size_t frame_bytes = (size_t)h->cfg.in_rows * h->cfg.in_cols * 3;
    const uint8_t* d_in = frames;
    if (!frames_on_device) {
        // Only the source rows k_pre reads cross the bus: the working image starts at row top_cutoff of the (resized)
        // frame, so the rows above its first source row are never touched (a third of a 640x480 frame with the
        // full-resolution geometry: the host-fed rate is PCIe bound).  One strided copy, the device layout stays
        // whole frames.
        int r0 = h->cfg.top_cutoff;
        if (h->pre.resize) r0 = dm::ifloor(h->cfg.top_cutoff * h->pre.ify) - 1;      // first row of the nearest-neighbour map, one row of slack
        r0 = r0 < 0 ? 0 : (r0 > h->cfg.in_rows - 1 ? h->cfg.in_rows - 1 : r0);
        const size_t skip = (size_t)r0 * h->cfg.in_cols * 3;
        LF_HIP_CHECK(h, hipMemcpy2DAsync(h->d_frames + skip, frame_bytes, frames + skip, frame_bytes, frame_bytes - skip, (size_t)n_frames,
                                         hipMemcpyHostToDevice, s));
        d_in = h->d_frames;
    }
    h->plugin_ready = false;
    int rc = h->detector == LF_DETECTOR_EDLINES ? run_detect_edlines(h, d_in, n_frames) : run_detect(h, d_in, n_frames, false);
    if (rc != LF_OK) return rc;
    rc = run_segments(h, n_frames, *out_dev, describe != 0);
    if (rc != LF_OK) return rc;
    // total + overflow flag travel to pinned host memory behind the kernels
    // one copy: overflow[0..3] -> h_pinned[1..4], the detector's failure count -> [5], the segment total (overflow[7]) -> [8]
    LF_HIP_CHECK(h, hipMemcpyAsync(&h->h_pinned[1], h->d_overflow, 8 * sizeof(int), hipMemcpyDeviceToHost, s));
    h->pending = true;
    h->pending_keylines = false;
    h->pending_problems = n_frames * 3;
    h->pending_capacity = out_dev->capacity;
    h->pend_in = d_in; h->pend_n = n_frames; h->pend_out = *out_dev; h->pend_describe = describe != 0;
    return LF_OK;
}

// The per-problem lists of the LSD stages hold LsdParams::rec_cap entries.  When a batch had a problem with more (d_overflow[5], read
// by the caller: the largest need), that problem was dropped on the device; here the lists are reallocated with room to spare and the
// caller runs the batch again.  The stream must be idle.  Results never depend on the capacity -- only whether a batch runs twice.
static int lsd_grow_lists(lf_handle* h, int need)
{
    size_t cap = ((size_t)need + (size_t)need / 4 + 4095) / 4096 * 4096;
    if (cap > h->Ps) cap = h->Ps;
    if (alloc_trace()) fprintf(stderr, "lanefront: a problem needs %d list entries, the handle holds %d: growing to %zu\n", need, h->lsd.rec_cap, cap);
    const int old_cap = h->lsd.rec_cap;
    free_lsd_lists(h);
    int rc = alloc_lsd_lists(h, (int)cap);
    if (rc == LF_OK) { ++h->lists_grown; h->lists_lost = false; return LF_OK; }
    // Out of memory part way: never leave the handle with null lists behind a capacity that says otherwise (the next batch would launch
    // the LSD kernels on them -- a GPU fault, not an error code).  Back to the capacity that did fit; when even that fails now, the
    // handle refuses every later detect call (lists_lost) until a growth succeeds.
    free_lsd_lists(h);
    if (alloc_lsd_lists(h, old_cap) != LF_OK) { free_lsd_lists(h); h->lsd.rec_cap = old_cap; h->lists_lost = true; }
    lf_set_error(h, LF_ERR_HIP, "out of device memory growing the LSD lists from %d to %zu entries per problem%s", old_cap, cap,
                 h->lists_lost ? "; the lists are gone: the handle refuses detection" : "; the handle keeps its old lists");
    return LF_ERR_HIP;
}

extern "C" int lf_wait(lf_handle* h, int* n_segments)
{
    if (!h) return LF_ERR_NOT_INITIALISED;
    LF_HIP_CHECK(h, hipSetDevice(h->device));
    LF_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    if (!h->pending) { if (n_segments) *n_segments = 0; return LF_OK; }
    h->pending = false;
    if (h->pending_keylines) {
        h->pending_keylines = false;
        int total_kl = 0, overflow = 0;
        keylines_pending_result(h, &total_kl, &overflow);
        if (n_segments) *n_segments = total_kl;
        if (overflow) { lf_set_error(h, LF_ERR_CAPACITY, "%d KeyLines exceed the output capacity %d", total_kl, h->pending_capacity); return LF_ERR_CAPACITY; }
        return LF_OK;
    }
    for (int attempt = 0; h->detector != LF_DETECTOR_EDLINES && h->h_pinned[6] > h->lsd.rec_cap; ++attempt) {
        // a problem did not fit the per-problem lists: grow them and run the batch again (its inputs are still where they were)
        if (attempt == 4) { lf_set_error(h, LF_ERR_CAPACITY, "the LSD lists keep overflowing (%d entries needed)", h->h_pinned[6]); return LF_ERR_CAPACITY; }
        int rc = lsd_grow_lists(h, h->h_pinned[6]);
        if (rc == LF_OK) rc = run_detect(h, h->pend_in, h->pend_n, false);
        if (rc == LF_OK) rc = run_segments(h, h->pend_n, h->pend_out, h->pend_describe);
        if (rc != LF_OK) return rc;
        LF_HIP_CHECK(h, hipMemcpyAsync(&h->h_pinned[1], h->d_overflow, 8 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        LF_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    }
    h->detector_failures = h->detector == LF_DETECTOR_EDLINES ? h->h_pinned[5] : 0;
    const int total = h->h_pinned[8];
    if (n_segments) *n_segments = total;
    // The next batch's region-growing slices (performance only: the results do not depend on them): 13 KB while nearly every
    // problem fits it (the synthetic lane frames), 28 KB when more than 5 % overflow it (real camera frames have two to three
    // times the edge pixels), 40 KB when more than 25 % overflow 28 KB.
    if (h->pending_problems > 0) {
        const int over_small = h->h_pinned[2], over_medium = h->h_pinned[3], np = h->pending_problems;
        h->grow_lds_level = over_medium * 4 > np ? 2 : (over_small * 20 > np ? 1 : 0);
        h->grow_mixed = (h->grow_lds_level == 0 ? over_small : over_medium) * 100 > np;
    }
    if (h->h_pinned[1]) { lf_set_error(h, LF_ERR_CAPACITY, "an LSD run produced more than max_lines_per_color=%d lines", h->cap_lines); return LF_ERR_CAPACITY; }
    if (total > h->pending_capacity) { lf_set_error(h, LF_ERR_CAPACITY, "%d segments exceed the output capacity %d", total, h->pending_capacity); return LF_ERR_CAPACITY; }
    return LF_OK;
}

extern "C" int lf_process_batch(lf_handle* h, const uint8_t* frames, int n_frames, int frames_on_device,
                                lf_segments* out, int out_on_device, int describe, int* n_segments)
{
    if (!h) return LF_ERR_NOT_INITIALISED;
    if (!out) { lf_set_error(h, LF_ERR_BAD_ARG, "lf_process_batch: null out"); return LF_ERR_BAD_ARG; }
    lf_segments dev = out_on_device ? *out : h->d_out;
    if (!out_on_device) {
        dev.capacity = out->capacity < h->out_capacity ? out->capacity : h->out_capacity;
        // only compute what the caller asked for
        if (!out->normals) dev.normals = nullptr;
        if (!out->color) dev.color = nullptr;
        if (!out->pixels_normalized) dev.pixels_normalized = nullptr;
        if (!out->ground) dev.ground = nullptr;
        if (!out->keep) dev.keep = nullptr;
        if (!out->desc) dev.desc = nullptr;
        if (!out->code) dev.code = nullptr;
        dev.frame_offset = h->d_frame_offset;
        if (describe && !out->lines) { lf_set_error(h, LF_ERR_BAD_ARG, "describe needs out->lines"); return LF_ERR_BAD_ARG; }
    }
    int rc = lf_process_batch_async(h, frames, n_frames, frames_on_device, &dev, describe);
    if (rc != LF_OK) return rc;
    int total = 0;
    rc = lf_wait(h, &total);
    if (n_segments) *n_segments = total;
    if (rc != LF_OK) return rc;
    hipStream_t s = h->stream;
    if (!out_on_device) {
        const size_t n = (size_t)total;
        if (out->frame_offset) LF_HIP_CHECK(h, hipMemcpyAsync(out->frame_offset, h->d_frame_offset, (n_frames + 1) * sizeof(int), hipMemcpyDeviceToHost, s));
        if (out->lines) LF_HIP_CHECK(h, hipMemcpyAsync(out->lines, dev.lines, n * 4 * sizeof(float), hipMemcpyDeviceToHost, s));
        if (out->normals) LF_HIP_CHECK(h, hipMemcpyAsync(out->normals, dev.normals, n * 2 * sizeof(float), hipMemcpyDeviceToHost, s));
        if (out->color) LF_HIP_CHECK(h, hipMemcpyAsync(out->color, dev.color, n, hipMemcpyDeviceToHost, s));
        if (out->pixels_normalized) LF_HIP_CHECK(h, hipMemcpyAsync(out->pixels_normalized, dev.pixels_normalized, n * 4 * sizeof(float), hipMemcpyDeviceToHost, s));
        if (out->ground) LF_HIP_CHECK(h, hipMemcpyAsync(out->ground, dev.ground, n * 4 * sizeof(double), hipMemcpyDeviceToHost, s));
        if (out->keep) LF_HIP_CHECK(h, hipMemcpyAsync(out->keep, dev.keep, n, hipMemcpyDeviceToHost, s));
        if (describe && out->desc) LF_HIP_CHECK(h, hipMemcpyAsync(out->desc, dev.desc, n * 72 * sizeof(float), hipMemcpyDeviceToHost, s));
        if (describe && out->code) LF_HIP_CHECK(h, hipMemcpyAsync(out->code, dev.code, n * 32, hipMemcpyDeviceToHost, s));
        LF_HIP_CHECK(h, hipStreamSynchronize(s));
    }
    return LF_OK;
}

constexpr int kPlugEager = 1024;     // segments fetched with the image (more than a frame has at the plugin's geometries)

// the caller's image -> pinned staging (it may reuse its buffer at once, np.copy in line_detector_lsd.py:136) -> the device,
// asynchronously: no synchronisation before the kernels
static int plugin_stage_image(lf_handle* h, const uint8_t* bgr, int rows, int cols, int row_stride_bytes)
{
    const size_t need = (size_t)rows * cols * 3;
    if (h->plug_in_bytes < need) {
        if (h->plug_in) (void)hipHostFree(h->plug_in);
        h->plug_in = nullptr; h->plug_in_bytes = 0;
        LF_HIP_CHECK(h, hipHostMalloc((void**)&h->plug_in, need));
        h->plug_in_bytes = need;
    }
    LF_HIP_CHECK(h, hipStreamSynchronize(h->stream));          // the previous image's copy out of the staging buffer (normally long done)
    for (int y = 0; y < rows; ++y) memcpy(h->plug_in + (size_t)y * cols * 3, bgr + (size_t)y * row_stride_bytes, (size_t)cols * 3);
    LF_HIP_CHECK(h, hipMemcpyAsync(h->d_frames, h->plug_in, need, hipMemcpyHostToDevice, h->stream));
    return LF_OK;
}

// queue the copies of everything lf_detect_lines returns behind the kernels; layout of plug_host:
// [lines eager x 16][normals64 eager x 16][centers eager x 8][3 mask images]
static int plugin_fetch_results(lf_handle* h)
{
    hipStream_t s = h->stream;
    const int eager = kPlugEager < 3 * h->cap_lines ? kPlugEager : 3 * h->cap_lines;
    const size_t need = (size_t)eager * 40 + 3 * h->P;
    if (h->plug_host_bytes < need) {
        if (h->plug_host) (void)hipHostFree(h->plug_host);
        h->plug_host = nullptr; h->plug_host_bytes = 0;
        LF_HIP_CHECK(h, hipHostMalloc((void**)&h->plug_host, need));
        h->plug_host_bytes = need;
    }
    h->plug_eager = eager;
    uint8_t* p = h->plug_host;
    LF_HIP_CHECK(h, hipMemcpyAsync(p, h->d_out.lines, (size_t)eager * 16, hipMemcpyDeviceToHost, s));
    LF_HIP_CHECK(h, hipMemcpyAsync(p + (size_t)eager * 16, h->d_normals64, (size_t)eager * 16, hipMemcpyDeviceToHost, s));
    LF_HIP_CHECK(h, hipMemcpyAsync(p + (size_t)eager * 32, h->d_centers, (size_t)eager * 8, hipMemcpyDeviceToHost, s));
    LF_HIP_CHECK(h, hipMemcpyAsync(p + (size_t)eager * 40, h->dbg_masks.p, 3 * h->P, hipMemcpyDeviceToHost, s));
    return LF_OK;
}

extern "C" int lf_set_image(lf_handle* h, const uint8_t* bgr, int rows, int cols, int row_stride_bytes)
{
    if (!h) return LF_ERR_NOT_INITIALISED;
    if (!bgr) { lf_set_error(h, LF_ERR_BAD_ARG, "lf_set_image: null image"); return LF_ERR_BAD_ARG; }
    if (rows != h->Hc || cols != h->W) { lf_set_error(h, LF_ERR_BAD_ARG, "lf_set_image: image is %dx%d, handle expects %dx%d", rows, cols, h->Hc, h->W); return LF_ERR_BAD_ARG; }
    if (row_stride_bytes < cols * 3) { lf_set_error(h, LF_ERR_BAD_ARG, "lf_set_image: row stride %d < %d", row_stride_bytes, cols * 3); return LF_ERR_BAD_ARG; }
    LF_HIP_CHECK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    h->plugin_ready = false;
    int rc = plugin_stage_image(h, bgr, rows, cols, row_stride_bytes);
    if (rc != LF_OK) return rc;
    for (int attempt = 0;; ++attempt) {
    rc = run_detect(h, h->d_frames, 1, true);
    if (rc != LF_OK) return rc;
    lf_segments dev = h->d_out;
    dev.desc = nullptr; dev.code = nullptr;
    rc = run_segments(h, 1, dev, false);
    if (rc != LF_OK) return rc;
    // Detections.area = the dilated colour mask as 0/255 bytes (line_detector_lsd.py:127-133): expanded once for the
    // three colours from the bit planes
    if ((rc = ensure(h, h->dbg_masks, 3 * h->P)) != LF_OK) return rc;
    launch_edges_u8(h->canny, 3, h->d_maskbits, (uint8_t*)h->dbg_masks.p, s);
    h->h_counts.resize(3); h->h_seg_offset.resize(4);
    if ((rc = plugin_fetch_results(h)) != LF_OK) return rc;
    LF_HIP_CHECK(h, hipMemcpyAsync(h->h_counts.data(), h->d_counts, 3 * sizeof(int), hipMemcpyDeviceToHost, s));
    LF_HIP_CHECK(h, hipMemcpyAsync(h->h_seg_offset.data(), h->d_seg_offset, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    LF_HIP_CHECK(h, hipMemcpyAsync(&h->h_pinned[6], h->d_overflow + 5, sizeof(int), hipMemcpyDeviceToHost, s));
    LF_HIP_CHECK(h, hipStreamSynchronize(s));          // the ONE synchronisation of an image: counts, segments and masks are on the host
    // (a handle made for batches holds short per-problem lists: grow them and detect again when this image needs more)
    if (h->h_pinned[6] <= h->lsd.rec_cap) break;
    if (attempt == 4) { lf_set_error(h, LF_ERR_CAPACITY, "the LSD lists keep overflowing (%d entries needed)", h->h_pinned[6]); return LF_ERR_CAPACITY; }
    if ((rc = lsd_grow_lists(h, h->h_pinned[6])) != LF_OK) return rc;
    }
    h->plugin_ready = true;
    return LF_OK;
}

extern "C" int lf_detect_lines(lf_handle* h, int color, float* lines4, double* normals2, float* centers2,
                               uint8_t* area_or_null, int cap, int* n_out)
{
    if (!h) return LF_ERR_NOT_INITIALISED;
    if (color < 0 || color > 2) { lf_set_error(h, LF_ERR_BAD_ARG, "Error: Undefined color strings..."); return LF_ERR_BAD_ARG; }
    if (!h->plugin_ready) { lf_set_error(h, LF_ERR_NOT_INITIALISED, "lf_detect_lines before lf_set_image"); return LF_ERR_NOT_INITIALISED; }
    if (!n_out) { lf_set_error(h, LF_ERR_BAD_ARG, "lf_detect_lines: n_out is null"); return LF_ERR_BAD_ARG; }
    LF_HIP_CHECK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    int n = h->h_counts[color];
    if (n > h->cap_lines) { lf_set_error(h, LF_ERR_CAPACITY, "LSD found %d lines, max_lines_per_color is %d", n, h->cap_lines); return LF_ERR_CAPACITY; }
    if (n > cap) { lf_set_error(h, LF_ERR_CAPACITY, "%d lines exceed caller capacity %d", n, cap); return LF_ERR_CAPACITY; }
    const size_t off = (size_t)h->h_seg_offset[color];
    if (h->plug_host && off + (size_t)n <= (size_t)h->plug_eager) {
        // everything came with the image: host copies, no device work and no synchronisation
        const uint8_t* p = h->plug_host;
        const size_t e = (size_t)h->plug_eager;
        if (n > 0) {
            if (lines4) memcpy(lines4, p + off * 16, (size_t)n * 16);
            if (normals2) memcpy(normals2, p + e * 16 + off * 16, (size_t)n * 16);
            if (centers2) memcpy(centers2, p + e * 32 + off * 8, (size_t)n * 8);
        }
        if (area_or_null) memcpy(area_or_null, p + e * 40 + (size_t)color * h->P, h->P);
        *n_out = n;
        return LF_OK;
    }
    if (n > 0) {
        if (lines4) LF_HIP_CHECK(h, hipMemcpyAsync(lines4, h->d_out.lines + off * 4, (size_t)n * 4 * sizeof(float), hipMemcpyDeviceToHost, s));
        if (normals2) LF_HIP_CHECK(h, hipMemcpyAsync(normals2, h->d_normals64 + off * 2, (size_t)n * 2 * sizeof(double), hipMemcpyDeviceToHost, s));
        if (centers2) LF_HIP_CHECK(h, hipMemcpyAsync(centers2, h->d_centers + off * 2, (size_t)n * 2 * sizeof(float), hipMemcpyDeviceToHost, s));
    }
    if (area_or_null) LF_HIP_CHECK(h, hipMemcpyAsync(area_or_null, static_cast<uint8_t*>(h->dbg_masks.p) + (size_t)color * h->P, h->P, hipMemcpyDeviceToHost, s));
    LF_HIP_CHECK(h, hipStreamSynchronize(s));
    *n_out = n;
    return LF_OK;
}

extern "C" int lf_associate(lf_handle* h, const uint8_t* query32, int nq, const uint8_t* map32, int nm,
                            int32_t* idx, float* dist, int on_device)
{
    if (!h) return LF_ERR_NOT_INITIALISED;
    if (nq < 0 || nm < 0 || (nq > 0 && (!query32 || !idx || !dist)) || (nm > 0 && !map32)) { lf_set_error(h, LF_ERR_BAD_ARG, "lf_associate: bad argument"); return LF_ERR_BAD_ARG; }
    if (nm > (1 << 21)) { lf_set_error(h, LF_ERR_UNSUPPORTED, "map larger than 2^21 entries"); return LF_ERR_UNSUPPORTED; }
    if (nq == 0) return LF_OK;
    LF_HIP_CHECK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    if (nm == 0) {
        // descriptor matrices cannot be void (binary_descriptor_matcher.cpp:201-205): report "no match"
        std::vector<int32_t> hi(nq, -1); std::vector<float> hd(nq, -1.f);
        if (on_device) {
            LF_HIP_CHECK(h, hipMemcpyAsync(idx, hi.data(), nq * sizeof(int32_t), hipMemcpyHostToDevice, s));
            LF_HIP_CHECK(h, hipMemcpyAsync(dist, hd.data(), nq * sizeof(float), hipMemcpyHostToDevice, s));
            LF_HIP_CHECK(h, hipStreamSynchronize(s));
        } else { memcpy(idx, hi.data(), nq * sizeof(int32_t)); memcpy(dist, hd.data(), nq * sizeof(float)); }
        return LF_OK;
    }
    const size_t nm_pad = assoc_rows_padded_m(nm);
    int rc;
    if ((rc = ensure(h, h->a_mx, nm_pad * 256)) || (rc = ensure(h, h->a_mcx, nm_pad * 32))) return rc;
    // the int8 A/B kernels (LF_ASSOC_INT8) have no tie pass: the lowest index there, said once
    const bool ties = h->tie_rule == LF_TIE_MIHASHER && !getenv("LF_ASSOC_INT8");
    if (h->tie_rule == LF_TIE_MIHASHER && !ties) { static bool told = false; if (!told) { told = true; fprintf(stderr, "lanefront: LF_ASSOC_INT8 is set: lf_associate falls back to LF_TIE_LOWEST (the int8 A/B kernels have no tie pass)\n"); } }
    if (ties && (rc = ensure(h, h->a_best, (size_t)nq * 8)) != LF_OK) return rc;
    const uint8_t *dq = query32, *dmp = map32;
    int32_t* didx = idx; float* ddist = dist;
    if (!on_device) {
        if ((rc = ensure(h, h->a_q, (size_t)nq * 32)) || (rc = ensure(h, h->a_m, (size_t)nm * 32)) ||
            (rc = ensure(h, h->a_idx, (size_t)nq * 4)) || (rc = ensure(h, h->a_dist, (size_t)nq * 4))) return rc;
        LF_HIP_CHECK(h, hipMemcpyAsync(h->a_q.p, query32, (size_t)nq * 32, hipMemcpyHostToDevice, s));
        LF_HIP_CHECK(h, hipMemcpyAsync(h->a_m.p, map32, (size_t)nm * 32, hipMemcpyHostToDevice, s));
        dq = (const uint8_t*)h->a_q.p; dmp = (const uint8_t*)h->a_m.p; didx = (int32_t*)h->a_idx.p; ddist = (float*)h->a_dist.p;
    }
    {
        StageTimer t(h, ST_ASSOC);
        h->a_ws.tie_res = ties ? static_cast<unsigned long long*>(h->a_best.p) : nullptr;      // (the distance pass then lists the queries of the tie pass)
        LF_HIP_CHECK(h, launch_assoc(dq, nq, dmp, nm, (int8_t*)h->a_mx.p, (int8_t*)h->a_mcx.p, h->a_ws, didx, ddist, s));
        if (ties)
            LF_HIP_CHECK(h, launch_assoc_ties(dq, nullptr, nq, (const int8_t*)h->a_mx.p, dmp, nullptr, nm, nullptr, 0, h->a_ws,
                                              static_cast<unsigned long long*>(h->a_best.p), didx, ddist, s));
    }
    LF_HIP_CHECK(h, hipGetLastError());
    if (!on_device) {
        LF_HIP_CHECK(h, hipMemcpyAsync(idx, didx, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
        LF_HIP_CHECK(h, hipMemcpyAsync(dist, ddist, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
        LF_HIP_CHECK(h, hipStreamSynchronize(s));
    }
    return LF_OK;
}

extern "C" int lf_set_tie_rule(lf_handle* h, int tie_rule)
{
    if (!h) return LF_ERR_NOT_INITIALISED;
    if (tie_rule != LF_TIE_LOWEST && tie_rule != LF_TIE_MIHASHER) { lf_set_error(h, LF_ERR_BAD_ARG, "lf_set_tie_rule: unknown rule %d", tie_rule); return LF_ERR_BAD_ARG; }
    h->tie_rule = tie_rule;
    return LF_OK;
}

extern "C" int lf_associate_float(lf_handle* h, const float* query72, int nq, const float* map72, int nm,
                                  int32_t* idx, float* dist, int on_device)
{
    if (!h) return LF_ERR_NOT_INITIALISED;
    if (nq <= 0 || nm <= 0 || !query72 || !map72 || !idx || !dist) { lf_set_error(h, LF_ERR_BAD_ARG, "lf_associate_float: bad argument"); return LF_ERR_BAD_ARG; }
    LF_HIP_CHECK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    int rc;
    if ((rc = ensure(h, h->a_best, (size_t)nq * 8)) || (rc = ensure(h, h->a_qn, (size_t)nq * 4)) || (rc = ensure(h, h->a_mn, (size_t)nm * 4))) return rc;
    const float *dq = query72, *dmp = map72;
    int32_t* didx = idx; float* ddist = dist;
    if (!on_device) {
        if ((rc = ensure(h, h->a_q, (size_t)nq * 288)) || (rc = ensure(h, h->a_m, (size_t)nm * 288)) ||
            (rc = ensure(h, h->a_idx, (size_t)nq * 4)) || (rc = ensure(h, h->a_dist, (size_t)nq * 4))) return rc;
        LF_HIP_CHECK(h, hipMemcpyAsync(h->a_q.p, query72, (size_t)nq * 288, hipMemcpyHostToDevice, s));
        LF_HIP_CHECK(h, hipMemcpyAsync(h->a_m.p, map72, (size_t)nm * 288, hipMemcpyHostToDevice, s));
        dq = (const float*)h->a_q.p; dmp = (const float*)h->a_m.p; didx = (int32_t*)h->a_idx.p; ddist = (float*)h->a_dist.p;
    }
    {
        StageTimer t(h, ST_ASSOC);
        launch_assoc_float(dq, nq, dmp, nm, (float*)h->a_qn.p, (float*)h->a_mn.p, (unsigned long long*)h->a_best.p, didx, ddist, s);
    }
    LF_HIP_CHECK(h, hipGetLastError());
    if (!on_device) {
        LF_HIP_CHECK(h, hipMemcpyAsync(idx, didx, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
        LF_HIP_CHECK(h, hipMemcpyAsync(dist, ddist, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
        LF_HIP_CHECK(h, hipStreamSynchronize(s));
    }
    return LF_OK;
}

// knnMatch / radiusMatch (binary_descriptor_matcher.cpp:258-335, 428-504): k_knn.hip
extern "C" int lf_select_queries(lf_handle* h, const uint8_t* query32, int nq, const uint8_t* mask, uint8_t* selected32, int32_t* query_idx,
                                 int* n_selected, int on_device)
{
    if (!h) return LF_ERR_NOT_INITIALISED;
    if (nq < 0 || !n_selected || (nq > 0 && (!query32 || !mask || !selected32 || !query_idx))) { lf_set_error(h, LF_ERR_BAD_ARG, "lf_select_queries: bad argument"); return LF_ERR_BAD_ARG; }
    *n_selected = 0;
    if (nq == 0) return LF_OK;
    LF_HIP_CHECK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    int rc;
    if ((rc = ensure(h, h->kn_total, 4 * sizeof(int))) != LF_OK) return rc;
    const uint8_t *dq = query32, *dmask = mask;
    uint8_t* dsel = selected32; int32_t* dqi = query_idx;
    if (!on_device) {
        if ((rc = ensure(h, h->a_q, (size_t)nq * 32)) || (rc = ensure(h, h->a_m, (size_t)nq * 33)) || (rc = ensure(h, h->a_idx, (size_t)nq * 4))) return rc;
        LF_HIP_CHECK(h, hipMemcpyAsync(h->a_q.p, query32, (size_t)nq * 32, hipMemcpyHostToDevice, s));
        uint8_t* m8 = static_cast<uint8_t*>(h->a_m.p) + (size_t)nq * 32;
        LF_HIP_CHECK(h, hipMemcpyAsync(m8, mask, (size_t)nq, hipMemcpyHostToDevice, s));
        dq = static_cast<const uint8_t*>(h->a_q.p); dmask = m8; dsel = static_cast<uint8_t*>(h->a_m.p); dqi = static_cast<int32_t*>(h->a_idx.p);
    }
    launch_select_queries(dq, dmask, nq, dsel, dqi, static_cast<int*>(h->kn_total.p), s);
    LF_HIP_CHECK(h, hipGetLastError());
    int n = 0;
    LF_HIP_CHECK(h, hipMemcpyAsync(&n, h->kn_total.p, sizeof(int), hipMemcpyDeviceToHost, s));
    LF_HIP_CHECK(h, hipStreamSynchronize(s));
    *n_selected = n;
    if (!on_device && n > 0) {
        LF_HIP_CHECK(h, hipMemcpyAsync(selected32, dsel, (size_t)n * 32, hipMemcpyDeviceToHost, s));
        LF_HIP_CHECK(h, hipMemcpyAsync(query_idx, dqi, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        LF_HIP_CHECK(h, hipStreamSynchronize(s));
    }
    return LF_OK;
}

extern "C" int lf_knn_match(lf_handle* h, const uint8_t* query32, int nq, const uint8_t* map32, int nm, int k, int32_t* idx, float* dist,
                            int on_device)
{
    if (!h) return LF_ERR_NOT_INITIALISED;
    if (nq < 0 || nm < 0 || k < 1 || k > 16 || (nq > 0 && (!query32 || !idx || !dist)) || (nm > 0 && !map32)) {
        lf_set_error(h, LF_ERR_BAD_ARG, "lf_knn_match: bad argument (k must be 1..16)");
        return LF_ERR_BAD_ARG;
    }
    if (nm > (1 << 24)) { lf_set_error(h, LF_ERR_UNSUPPORTED, "lf_knn_match: map larger than 2^24 entries"); return LF_ERR_UNSUPPORTED; }
    if (nq == 0) return LF_OK;
    LF_HIP_CHECK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    int rc;
    const uint8_t *dq = query32, *dm_ = map32;
    int32_t* didx = idx; float* ddist = dist;
    const size_t out = (size_t)nq * k;
    if (!on_device) {
        if ((rc = ensure(h, h->a_q, (size_t)nq * 32)) || (rc = ensure(h, h->a_m, (size_t)(nm > 0 ? nm : 1) * 32)) ||
            (rc = ensure(h, h->a_idx, out * 4)) || (rc = ensure(h, h->a_dist, out * 4))) return rc;
        LF_HIP_CHECK(h, hipMemcpyAsync(h->a_q.p, query32, (size_t)nq * 32, hipMemcpyHostToDevice, s));
        if (nm > 0) LF_HIP_CHECK(h, hipMemcpyAsync(h->a_m.p, map32, (size_t)nm * 32, hipMemcpyHostToDevice, s));
        dq = (const uint8_t*)h->a_q.p; dm_ = (const uint8_t*)h->a_m.p; didx = (int32_t*)h->a_idx.p; ddist = (float*)h->a_dist.p;
    }
    { StageTimer t(h, ST_ASSOC); launch_knn(dq, nq, dm_, nm, k, 128, h->tie_rule == LF_TIE_MIHASHER, didx, ddist, s); }
    LF_HIP_CHECK(h, hipGetLastError());
    if (!on_device) {
        LF_HIP_CHECK(h, hipMemcpyAsync(idx, didx, out * 4, hipMemcpyDeviceToHost, s));
        LF_HIP_CHECK(h, hipMemcpyAsync(dist, ddist, out * 4, hipMemcpyDeviceToHost, s));
        LF_HIP_CHECK(h, hipStreamSynchronize(s));
    }
    return LF_OK;
}

extern "C" int lf_radius_match(lf_handle* h, const uint8_t* query32, int nq, const uint8_t* map32, int nm, float max_distance,
                               int32_t* offsets, int32_t* idx, float* dist, int cap, int* total_out, int on_device)
{
    if (!h) return LF_ERR_NOT_INITIALISED;
    if (nq < 0 || nm < 0 || cap < 0 || !offsets || (cap > 0 && (!idx || !dist)) || (nq > 0 && !query32) || (nm > 0 && !map32) || !(max_distance >= 0)) {
        lf_set_error(h, LF_ERR_BAD_ARG, "lf_radius_match: bad argument");
        return LF_ERR_BAD_ARG;
    }
    if (nm > (1 << 24)) { lf_set_error(h, LF_ERR_UNSUPPORTED, "lf_radius_match: map larger than 2^24 entries"); return LF_ERR_UNSUPPORTED; }
    // K = N results are only ever collected up to D = 128 bits (Mihasher, :721), then filtered by maxDistance (:474)
    int md = max_distance >= 128.f ? 128 : (int)max_distance;
    LF_HIP_CHECK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    if (nq == 0) { if (on_device) LF_HIP_CHECK(h, hipMemsetAsync(offsets, 0, sizeof(int32_t), s)); else offsets[0] = 0; if (total_out) *total_out = 0; return LF_OK; }
    int rc;
    if ((rc = ensure(h, h->kn_hist, (size_t)nq * 129 * 4)) || (rc = ensure(h, h->kn_count, (size_t)nq * 4)) || (rc = ensure(h, h->kn_off, (size_t)(nq + 1) * 4)) ||
        (rc = ensure(h, h->kn_total, 16))) return rc;
    const uint8_t *dq = query32, *dm_ = map32;
    int32_t *doff = offsets, *didx = idx; float* ddist = dist;
    if (!on_device) {
        if ((rc = ensure(h, h->a_q, (size_t)nq * 32)) || (rc = ensure(h, h->a_m, (size_t)(nm > 0 ? nm : 1) * 32)) ||
            (rc = ensure(h, h->a_idx, (size_t)(cap > 0 ? cap : 1) * 4)) || (rc = ensure(h, h->a_dist, (size_t)(cap > 0 ? cap : 1) * 4))) return rc;
        LF_HIP_CHECK(h, hipMemcpyAsync(h->a_q.p, query32, (size_t)nq * 32, hipMemcpyHostToDevice, s));
        if (nm > 0) LF_HIP_CHECK(h, hipMemcpyAsync(h->a_m.p, map32, (size_t)nm * 32, hipMemcpyHostToDevice, s));
        dq = (const uint8_t*)h->a_q.p; dm_ = (const uint8_t*)h->a_m.p; doff = (int32_t*)h->kn_off.p; didx = (int32_t*)h->a_idx.p; ddist = (float*)h->a_dist.p;
    }
    {
        StageTimer t(h, ST_ASSOC);
        launch_radius(dq, nq, dm_, nm, md, (int32_t*)h->kn_hist.p, (int32_t*)h->kn_count.p, doff, (int*)h->kn_total.p, cap, h->tie_rule == LF_TIE_MIHASHER, didx, ddist, s);
    }
    LF_HIP_CHECK(h, hipGetLastError());
    int total = 0;
    LF_HIP_CHECK(h, hipMemcpyAsync(&total, h->kn_total.p, sizeof(int), hipMemcpyDeviceToHost, s));
    if (!on_device) LF_HIP_CHECK(h, hipMemcpyAsync(offsets, doff, (size_t)(nq + 1) * 4, hipMemcpyDeviceToHost, s));
    LF_HIP_CHECK(h, hipStreamSynchronize(s));
    if (total_out) *total_out = total;
    if (total > cap) { lf_set_error(h, LF_ERR_CAPACITY, "lf_radius_match: %d matches exceed the capacity %d (offsets are complete: size the arrays from them)", total, cap); return LF_ERR_CAPACITY; }
    if (!on_device && total > 0) {
        LF_HIP_CHECK(h, hipMemcpyAsync(idx, didx, (size_t)total * 4, hipMemcpyDeviceToHost, s));
        LF_HIP_CHECK(h, hipMemcpyAsync(dist, ddist, (size_t)total * 4, hipMemcpyDeviceToHost, s));
        LF_HIP_CHECK(h, hipStreamSynchronize(s));
    }
    return LF_OK;
}

// LSD alone on a caller-supplied binary image (any non-zero byte = edge pixel): the LSD stages
// of the pipeline (gradient -> order -> grow) with the colour mask forced to all ones.  Test and
// diagnosis entry; lines are in working-image pixels before normal-based reordering, exactly what
// cv2's detect() would return for this image under the oracle's restatement.
// anti-instagram colour clustering (k_kmeans.hip): kmeans.py:22-47
extern "C" int lf_kmeans(lf_handle* h, const uint8_t* bgr_points, int n, int on_device, int k, const double* init_centers, int max_iter,
                         double tol, double* centers_out, long long* counts_out, double* inertia_out, int* n_iter_out)
{
    if (!h) return LF_ERR_NOT_INITIALISED;
    if (!bgr_points || !init_centers || !centers_out || !counts_out || n < 1 || k < 1 || k > 16 || max_iter < 1) {
        lf_set_error(h, LF_ERR_BAD_ARG, "lf_kmeans: null argument, n < 1, max_iter < 1 or k outside 1..16");
        return LF_ERR_BAD_ARG;
    }
    if (n > (1 << 24)) {      // k_kmeans' per-wave 32-bit colour sums (64 lanes x n / 1024 points x 255) stay exact up to here
        lf_set_error(h, LF_ERR_UNSUPPORTED, "lf_kmeans: more than 2^24 points (%d) are not supported", n);
        return LF_ERR_UNSUPPORTED;
    }
    LF_HIP_CHECK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    int rc;
    // f64 scratch: [0 .. 3k) init, [64 .. 64 + 3k] centres + inertia; counts: [k] + the status word behind them
    if ((rc = ensure(h, h->km_lab, (size_t)n)) || (rc = ensure(h, h->km_f64, 128 * sizeof(double))) || (rc = ensure(h, h->km_cnt, 32 * sizeof(long long)))) return rc;
    const uint8_t* dp = bgr_points;
    if (!on_device) {
        if ((rc = ensure(h, h->km_pts, (size_t)n * 3)) != LF_OK) return rc;
        LF_HIP_CHECK(h, hipMemcpyAsync(h->km_pts.p, bgr_points, (size_t)n * 3, hipMemcpyHostToDevice, s));
        dp = static_cast<const uint8_t*>(h->km_pts.p);
    }
    double* f64 = static_cast<double*>(h->km_f64.p);
    long long* cnt = static_cast<long long*>(h->km_cnt.p);
    int* status = reinterpret_cast<int*>(cnt + 16);
    LF_HIP_CHECK(h, hipMemcpyAsync(f64, init_centers, (size_t)k * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    launch_kmeans(dp, n, k, f64, max_iter, tol, static_cast<uint8_t*>(h->km_lab.p), f64 + 64, cnt, status, s);
    LF_HIP_CHECK(h, hipGetLastError());
    double res[49];
    long long hc[17];
    LF_HIP_CHECK(h, hipMemcpyAsync(res, f64 + 64, (size_t)(3 * k + 1) * sizeof(double), hipMemcpyDeviceToHost, s));
    LF_HIP_CHECK(h, hipMemcpyAsync(hc, cnt, 17 * sizeof(long long), hipMemcpyDeviceToHost, s));
    LF_HIP_CHECK(h, hipStreamSynchronize(s));
    const int iters = *reinterpret_cast<int*>(&hc[16]);
    if (iters < 0) { lf_set_error(h, LF_ERR_BAD_ARG, "lf_kmeans: a cluster stayed empty (fewer distinct samples than clusters)"); return LF_ERR_BAD_ARG; }
    for (int j = 0; j < 3 * k; ++j) centers_out[j] = res[j];
    for (int j = 0; j < k; ++j) counts_out[j] = hc[j];
    if (inertia_out) *inertia_out = res[3 * k];
    if (n_iter_out) *n_iter_out = iters;
    return LF_OK;
}

extern "C" int lf_lsd_list_capacity(const lf_handle* h, int* entries, int* grown)
{
    if (!h) return LF_ERR_NOT_INITIALISED;
    if (entries) *entries = h->lsd.rec_cap;
    if (grown) *grown = h->lists_grown;
    return LF_OK;
}

extern "C" int lf_lsd_scratch_stride(const lf_handle* h)
{
    if (!h) return LF_ERR_NOT_INITIALISED;
    return (int)lsd_grow_reg_stride(h->lsd);
}

extern "C" int lf_suggested_depth(const lf_handle* h)
{
    if (!h) return LF_ERR_NOT_INITIALISED;
    return h->grow_lds_level == 0 && !h->grow_mixed ? 8 : 18;
}

extern "C" int lf_debug_std_sort(lf_handle* h, const int32_t* keys, int n, int32_t* order)
{
    if (!h) return LF_ERR_NOT_INITIALISED;
    if (!keys || !order || n < 1 || n >= (1 << 20)) { lf_set_error(h, LF_ERR_BAD_ARG, "lf_debug_std_sort: bad argument (1 <= n < 2^20)"); return LF_ERR_BAD_ARG; }
    LF_HIP_CHECK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    std::vector<uint32_t> e((size_t)n);
    for (int i = 0; i < n; ++i) {
        if (keys[i] < 0 || keys[i] > 1023) { lf_set_error(h, LF_ERR_BAD_ARG, "lf_debug_std_sort: keys must be in [0, 1023]"); return LF_ERR_BAD_ARG; }
        e[i] = ((uint32_t)keys[i] << 20) | (uint32_t)(i + 1);
    }
    int rc;
    const size_t words = std_sort_debug_words(n);
    if ((rc = ensure(h, h->a_q, (size_t)n * 4)) || (rc = ensure(h, h->a_m, words * 4)) || (rc = ensure(h, h->a_best, 64))) return rc;
    LF_HIP_CHECK(h, hipMemcpyAsync(h->a_q.p, e.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
    launch_std_sort_debug(static_cast<const uint32_t*>(h->a_q.p), static_cast<uint32_t*>(h->a_m.p), n, static_cast<int*>(h->a_best.p), s);
    LF_HIP_CHECK(h, hipGetLastError());
    int cnt = 0;
    LF_HIP_CHECK(h, hipMemcpyAsync(&cnt, h->a_best.p, sizeof(int), hipMemcpyDeviceToHost, s));
    LF_HIP_CHECK(h, hipStreamSynchronize(s));
    // the sorted non-zero keys sit in the `out` area of the work buffer (k_lsd_seed32.hip: seed_work): 4 * cap words in
    const size_t cap = words / 12;
    if (cnt > 0) LF_HIP_CHECK(h, hipMemcpy(e.data(), static_cast<uint32_t*>(h->a_m.p) + 4 * cap, (size_t)cnt * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < cnt; ++i) order[i] = (int32_t)(e[i] & 0xfffffu);
    // the elements with key 0 (the detector's flat pixels) are anonymous: listed behind, by index
    {
        int k = cnt;
        for (int i = 0; i < n; ++i) if (keys[i] == 0) order[k++] = i;
    }
    return LF_OK;
}

extern "C" int lf_debug_lsd_binary(lf_handle* h, const uint8_t* img, int rows, int cols, float* lines4, int cap, int* n_out)
{
    if (!h) return LF_ERR_NOT_INITIALISED;
    if (!img || !lines4 || !n_out || rows != h->Hc || cols != h->W) { lf_set_error(h, LF_ERR_BAD_ARG, "lf_debug_lsd_binary: bad argument (image must be %dx%d)", h->Hc, h->W); return LF_ERR_BAD_ARG; }
    LF_HIP_CHECK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    if ((size_t)h->lsd.rec_cap < h->Ps) {                 // (a debug entry: any binary image must fit -- whole-image lists from here on)
        LF_HIP_CHECK(h, hipStreamSynchronize(s));
        const int rc = lsd_grow_lists(h, (int)h->Ps);
        if (rc != LF_OK) return rc;
    }
    const size_t nw = (size_t)h->Hc * h->Ww;
    std::vector<uint32_t> bits(nw, 0u), ones(nw * 3, 0xffffffffu);
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x)
            if (img[(size_t)y * cols + x]) bits[(size_t)y * h->Ww + (x >> 5)] |= 1u << (x & 31);
    LF_HIP_CHECK(h, hipMemcpyAsync(h->d_strong, bits.data(), nw * 4, hipMemcpyHostToDevice, s));
    LF_HIP_CHECK(h, hipMemcpyAsync(h->d_maskbits, ones.data(), nw * 12, hipMemcpyHostToDevice, s));
    LF_HIP_CHECK(h, hipMemsetAsync(h->d_maxgrad, 0, 3 * sizeof(unsigned long long), s));
    launch_lsd_grad(h->lsd, h->rt, 1, h->d_strong, h->d_maskbits, h->d_raddr, h->d_rdeg, h->d_rmod, h->d_rcs, h->d_rsn, h->d_nrec, h->d_maxgrad,
                    h->max_nsx, h->max_nsy, h->d_tile_list, h->d_tile_count, h->d_laddr, h->d_lmod, h->d_nlow, h->d_overflow + 5, false, s);
    launch_lsd_order(h->lsd, 1, h->d_raddr, h->d_rdeg, h->d_rmod, h->d_rcs, h->d_rsn, h->d_nrec, h->d_maxgrad, h->d_sort_a, h->d_sort_b, h->d_order_a, h->d_order_b, h->d_norder,
                     h->d_cxy, h->d_cdeg, h->d_cmod, h->d_ccs, h->d_csn, h->d_row_start, s);
    if (h->cfg.lsd_seed_order == LF_LSD_SEED_OPENCV32)
        launch_lsd_seed32(h->lsd, 1, h->d_nrec, h->d_norder, h->d_overflow + 5, h->d_maxgrad, h->d_cxy, h->d_cmod, h->d_laddr, h->d_lmod, h->d_nlow, h->d_sort_a, h->d_sort_b, h->d_order_a, h->d_order_b, 0, s);
    launch_lsd_label(h->lsd, 1, h->d_norder, h->d_cxy, h->d_row_start, h->d_clabel, h->d_comp_list, h->d_comp_count, h->d_comp_key, h->d_reg, s);
    launch_lsd_grow(h->lsd, 1, h->d_order_a, h->d_norder, h->d_cxy, h->d_cdeg, h->d_cmod, h->d_ccs, h->d_csn, h->d_row_start,
                    h->d_clabel, h->d_comp_list, h->d_comp_count, kCompCap, h->d_reg, h->d_gused, h->d_tmp_lines, h->d_tmp_tags,
                    h->d_slot_lines, h->d_counts, nullptr, h->d_pend_rec, h->d_pend_tag, h->d_pend_count, kGrowLdsKb[h->grow_lds_level], true,
                    h->env_bitmap, s);
    LF_HIP_CHECK(h, hipGetLastError());
    int n = 0;
    LF_HIP_CHECK(h, hipMemcpyAsync(&n, h->d_counts, sizeof(int), hipMemcpyDeviceToHost, s));
    LF_HIP_CHECK(h, hipStreamSynchronize(s));
    h->last_frames = 1;
    h->plugin_ready = false;
    *n_out = n;
    if (n > h->cap_lines) { lf_set_error(h, LF_ERR_CAPACITY, "LSD found %d lines, max_lines_per_color is %d", n, h->cap_lines); return LF_ERR_CAPACITY; }
    if (n > cap) { lf_set_error(h, LF_ERR_CAPACITY, "%d lines exceed caller capacity %d", n, cap); return LF_ERR_CAPACITY; }
    if (n > 0) LF_HIP_CHECK(h, hipMemcpy(lines4, h->d_slot_lines, (size_t)n * 4 * sizeof(float), hipMemcpyDeviceToHost));
    return LF_OK;
}

extern "C" int lf_lsd_size(const lf_handle* h, int* rows, int* cols)
{
    if (!h || !rows || !cols) return LF_ERR_BAD_ARG;
    *rows = h->Hs; *cols = h->Ws;
    return LF_OK;
}

extern "C" int lf_debug_fetch(lf_handle* h, int buffer_id, void* dst, size_t bytes)
{
    if (!h) return LF_ERR_NOT_INITIALISED;
    if (!dst) { lf_set_error(h, LF_ERR_BAD_ARG, "lf_debug_fetch: null dst"); return LF_ERR_BAD_ARG; }
    LF_HIP_CHECK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const size_t n = (size_t)h->last_frames;
    const void* src = nullptr;
    size_t avail = 0;
    switch (buffer_id) {
    case LF_BUF_BGR: {
        int rc = ensure(h, h->dbg_bgr, n * h->P * 3);
        if (rc != LF_OK) return rc;
        launch_bgrx_to_bgr((int)(n * h->P), h->d_bgr, (uint8_t*)h->dbg_bgr.p, s);
        src = h->dbg_bgr.p; avail = n * h->P * 3; break;
    }
    case LF_BUF_MASKS: {
        int rc = ensure(h, h->dbg_masks, n * 3 * h->P);
        if (rc != LF_OK) return rc;
        launch_edges_u8(h->canny, (int)(n * 3), h->d_maskbits, (uint8_t*)h->dbg_masks.p, s);     // same bit-plane layout as the edge map
        src = h->dbg_masks.p; avail = n * 3 * h->P; break;
    }
    case LF_BUF_EDGES:
        launch_edges_u8(h->canny, (int)n, h->d_strong, h->d_edges_u8, s);
        src = h->d_edges_u8; avail = n * h->P; break;
    case LF_BUF_LSD_ANGLE:
    case LF_BUF_LSD_MODGRAD: {
        // the pipeline keeps no dense LSD planes: rebuild them from the compact arrays
        int rc = ensure(h, h->dbg_ang, n * 3 * h->Ps * sizeof(float));
        if (rc == LF_OK) rc = ensure(h, h->dbg_mod, n * 3 * h->Ps * sizeof(double));
        if (rc != LF_OK) return rc;
        launch_lsd_dense_debug(h->lsd, (int)n, h->d_norder, h->d_cxy, h->d_cdeg, h->d_cmod, (float*)h->dbg_ang.p, (double*)h->dbg_mod.p, s);
        if (buffer_id == LF_BUF_LSD_ANGLE) { src = h->dbg_ang.p; avail = n * 3 * h->Ps * sizeof(float); }
        else { src = h->dbg_mod.p; avail = n * 3 * h->Ps * sizeof(double); }
        break;
    }
    case LF_BUF_LSD_ORDER: {
        // [frames][3][Hs * Ws] for the caller; the handle's lists have rec_cap entries per problem
        const size_t row = h->Ps * sizeof(uint32_t), have = (size_t)h->lsd.rec_cap * sizeof(uint32_t);
        if (bytes > n * 3 * row) { lf_set_error(h, LF_ERR_CAPACITY, "buffer %d holds %zu bytes, %zu requested", buffer_id, n * 3 * row, bytes); return LF_ERR_CAPACITY; }
        LF_HIP_CHECK(h, hipMemcpy2DAsync(dst, row, h->d_order_a, have, have, bytes / row, hipMemcpyDeviceToHost, s));
        LF_HIP_CHECK(h, hipStreamSynchronize(s));
        return LF_OK;
    }
    case LF_BUF_LSD_NORDER: src = h->d_norder; avail = n * 3 * sizeof(int); break;
    case LF_BUF_LBD_DX:
    case LF_BUF_LBD_DY: {
        // the pipeline keeps dx and dy interleaved: split them for the caller
        int rc = ensure(h, h->dbg_dx, n * h->P * sizeof(int16_t));
        if (rc == LF_OK) rc = ensure(h, h->dbg_dy, n * h->P * sizeof(int16_t));
        if (rc != LF_OK) return rc;
        launch_lbd_split_debug(n * h->P, h->d_dxy, (int16_t*)h->dbg_dx.p, (int16_t*)h->dbg_dy.p, s);
        src = buffer_id == LF_BUF_LBD_DX ? h->dbg_dx.p : h->dbg_dy.p;
        avail = n * h->P * sizeof(int16_t);
        break;
    }
    case LF_BUF_LSD_COUNTS: src = h->d_counts; avail = n * 3 * sizeof(int); break;
    case LF_BUF_LSD_NLOW:
        if (!h->d_nlow) { memset(dst, 0, bytes < n * 3 * sizeof(int) ? bytes : n * 3 * sizeof(int)); return LF_OK; }
        src = h->d_nlow; avail = n * 3 * sizeof(int); break;
    case LF_BUF_LSD_SCRATCH: src = h->d_reg; avail = n * 3 * lsd_grow_reg_stride(h->lsd) * sizeof(uint32_t); break;
    default: lf_set_error(h, LF_ERR_BAD_ARG, "unknown buffer id %d", buffer_id); return LF_ERR_BAD_ARG;
    }
    if (bytes > avail) { lf_set_error(h, LF_ERR_CAPACITY, "buffer %d holds %zu bytes, %zu requested", buffer_id, avail, bytes); return LF_ERR_CAPACITY; }
    LF_HIP_CHECK(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s));
    LF_HIP_CHECK(h, hipStreamSynchronize(s));
    return LF_OK;
}

extern "C" int lf_set_profiling(lf_handle* h, int enabled)
{
    if (!h) return LF_ERR_NOT_INITIALISED;
    h->profiling = enabled != 0;
    // the event pool is filled HERE, not inside the first profiled batches (an event pair per stage and batch: a 200-step run
    // would otherwise create some thousand events while it is being timed)
    if (h->profiling) {
        (void)hipSetDevice(h->device);
        while (h->ev_free.size() < 1024) {
            EvPair n; n.st = 0;
            if (hipEventCreate(&n.a) != hipSuccess) break;
            if (hipEventCreate(&n.b) != hipSuccess) { (void)hipEventDestroy(n.a); break; }
            h->ev_free.push_back(n);
        }
    }
    return LF_OK;
}

extern "C" int lf_reset_timing(lf_handle* h)
{
    if (!h) return LF_ERR_NOT_INITIALISED;
    timing_resolve(h);
    memset(h->ms, 0, sizeof(h->ms)); memset(h->launches, 0, sizeof(h->launches));
    return LF_OK;
}

extern "C" int lf_get_timing(lf_handle* h, double* ms_per_stage, int32_t* launches_per_stage, int n)
{
    if (!h) return LF_ERR_NOT_INITIALISED;
    timing_resolve(h);
    for (int i = 0; i < n && i < LF_N_STAGES; ++i) {
        if (ms_per_stage) ms_per_stage[i] = h->ms[i];
        if (launches_per_stage) launches_per_stage[i] = h->launches[i];
    }
    return LF_OK;
}

// ---------------------------------------------------------------------------------------- JPEG ingest
extern "C" int lf_jpeg_info(const uint8_t* jpeg, size_t jpeg_size, int* rows, int* cols, int* components, int* hmax, int* vmax)
{
    if (!jpeg) return LF_ERR_BAD_ARG;
    return lf::jpeg::peek(jpeg, jpeg_size, rows, cols, components, hmax, vmax);
}

extern "C" int lf_frames_buffer(lf_handle* h, uint8_t** device_ptr, size_t* bytes)
{
    if (!h) return LF_ERR_NOT_INITIALISED;
    if (device_ptr) *device_ptr = h->d_frames;
    if (bytes) *bytes = h->frames_bytes;
    return LF_OK;
}

extern "C" int lf_jpeg_decode_batch(lf_handle* h, const uint8_t* const* jpeg, const size_t* jpeg_size, int n_frames,
                                    int rows, int cols, uint8_t* frames, int frames_on_device, int n_threads,
                                    int* frame_status)
{
    if (!h) return LF_ERR_NOT_INITIALISED;
    if (!jpeg || !jpeg_size || !frames || n_frames < 1 || rows < 1 || cols < 1 || rows > 65535 || cols > 65535) {
        lf_set_error(h, LF_ERR_BAD_ARG, "lf_jpeg_decode_batch: null argument, n_frames < 1 or bad size %dx%d", rows, cols);
        return LF_ERR_BAD_ARG;
    }
    if (n_frames > 65535) { lf_set_error(h, LF_ERR_CAPACITY, "lf_jpeg_decode_batch: at most 65535 frames per call"); return LF_ERR_CAPACITY; }
    if (frames_on_device && frames == h->d_frames &&
        (size_t)n_frames * rows * cols * 3 > h->frames_bytes) {
        lf_set_error(h, LF_ERR_CAPACITY, "lf_jpeg_decode_batch: %d frames of %dx%d do not fit the handle's frame buffer (%d of %dx%d)",
                     n_frames, rows, cols, h->max_frames, h->cfg.in_rows, h->cfg.in_cols);
        return LF_ERR_CAPACITY;
    }
    LF_HIP_CHECK(h, hipSetDevice(h->device));
    if (!h->jpeg) {
        h->jpeg = new (std::nothrow) JpegState();
        if (!h->jpeg) { lf_set_error(h, LF_ERR_HIP, "out of host memory"); return LF_ERR_HIP; }
        LF_HIP_CHECK(h, hipEventCreateWithFlags(&h->jpeg->staged, hipEventDisableTiming));
    }
    JpegState& J = *h->jpeg;
    hipStream_t s = h->stream;
    static const bool trace = getenv("LF_JPEG_TRACE") != nullptr;       // diagnostic: per-phase host times on stderr
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    const auto t_begin = now();
    double t_decode = 0, t_wait = 0, t_pack = 0;
    if ((int)J.frames.size() < n_frames) J.frames.resize((size_t)n_frames);

    // ---- host: entropy decoding, one frame per task
    int nt = n_threads > 0 ? n_threads : (n_frames < 64 ? n_frames : 64);
    if (nt > n_frames) nt = n_frames;
    {
        std::atomic<int> next(0);
        J.pool.run(nt, [&](int) {
            for (;;) {
                const int i = next.fetch_add(1);
                if (i >= n_frames) break;
                lf::jpeg::FrameCoefs& fc = J.frames[(size_t)i];
                if (!jpeg[i]) { fc.status = LF_ERR_BAD_ARG; fc.hdr.valid = 0; fc.hdr.nblocks = 0; fc.n_entries = 0; continue; }
                // a stream of another size than the batch was declared with is refused right after its headers
                // (LF_ERR_BAD_ARG), before any host buffer is sized from the stream's own fields
                (void)lf::jpeg::decode_coefficients(jpeg[i], jpeg_size[i], fc, rows, cols);
                if (fc.status != LF_OK) { fc.hdr.valid = 0; fc.hdr.nblocks = 0; fc.n_entries = 0; }
            }
        });
    }
    t_decode = ms_since(t_begin);
    // ---- layout of the batch
    size_t total_entries = 0, total_blocks = 0;
    int max_blocks = 0, n_failed = 0, first_error = LF_OK;
    for (int i = 0; i < n_frames; ++i) {
        lf::jpeg::FrameCoefs& fc = J.frames[(size_t)i];
        if (frame_status) frame_status[i] = fc.status;
        if (fc.status != LF_OK) { ++n_failed; if (first_error == LF_OK) first_error = fc.status; }
        fc.hdr.entry_base = (uint32_t)total_entries;
        fc.hdr.block_base = (uint32_t)total_blocks;
        total_entries += fc.n_entries;
        total_blocks += (size_t)fc.hdr.nblocks;
        if (fc.hdr.nblocks > max_blocks) max_blocks = fc.hdr.nblocks;
    }
    if (total_entries >= (1ull << 32) || total_blocks >= (1ull << 32)) { lf_set_error(h, LF_ERR_CAPACITY, "batch too large"); return LF_ERR_CAPACITY; }
    const size_t hdr_bytes = (size_t)n_frames * sizeof(lf::jpeg::FrameHeader);
    const size_t blk_bytes = (total_blocks + 1) * sizeof(uint32_t);
    const size_t ent_bytes = (total_entries + 1) * sizeof(uint32_t);
    const size_t off_blk = (hdr_bytes + 255) & ~(size_t)255, off_ent = (off_blk + blk_bytes + 255) & ~(size_t)255;
    const size_t stage_bytes = off_ent + ent_bytes;
    // the previous call's copy out of the staging buffer must have completed before it is rewritten
    const auto t_w = now();
    if (J.staged_pending) { LF_HIP_CHECK(h, hipEventSynchronize(J.staged)); J.staged_pending = false; }
    t_wait = ms_since(t_w);
    if (J.h_stage_bytes < stage_bytes) {
        if (J.h_stage) (void)hipHostFree(J.h_stage);
        J.h_stage = nullptr; J.h_stage_bytes = 0;
        const size_t want = stage_bytes + stage_bytes / 4 + 4096;
        LF_HIP_CHECK(h, hipHostMalloc(&J.h_stage, want, hipHostMallocDefault));
        J.h_stage_bytes = want;
    }
    const auto t_p = now();
    {
        // pack headers | block ends | entries into the pinned staging buffer, frames in parallel
        uint8_t* st = static_cast<uint8_t*>(J.h_stage);
        std::atomic<int> next(0);
        J.pool.run(nt, [&](int) {
            for (;;) {
                const int i = next.fetch_add(1);
                if (i >= n_frames) break;
                const lf::jpeg::FrameCoefs& fc = J.frames[(size_t)i];
                memcpy(st + (size_t)i * sizeof(lf::jpeg::FrameHeader), &fc.hdr, sizeof(lf::jpeg::FrameHeader));
                if (fc.hdr.nblocks) memcpy(st + off_blk + (size_t)fc.hdr.block_base * 4, fc.block_end.data(), (size_t)fc.hdr.nblocks * 4);
                if (fc.n_entries) memcpy(st + off_ent + (size_t)fc.hdr.entry_base * 4, fc.entries.data(), fc.n_entries * 4);
            }
        });
    }
    t_pack = ms_since(t_p);
    // ---- device
    JpegGeom g;
    g.rows = rows; g.cols = cols;
    g.Wp = (cols + 15) / 16 * 16;
    g.Hp = (rows + 15) / 16 * 16;
    int rc;
    if ((rc = ensure(h, J.planes, (size_t)n_frames * 3 * g.Wp * g.Hp)) != LF_OK) return rc;
    if ((rc = ensure(h, J.hdrs, stage_bytes)) != LF_OK) return rc;       // one device image of the staging buffer
    uint8_t* d_stage = static_cast<uint8_t*>(J.hdrs.p);
    LF_HIP_CHECK(h, hipMemcpyAsync(d_stage, J.h_stage, stage_bytes, hipMemcpyHostToDevice, s));
    LF_HIP_CHECK(h, hipEventRecord(J.staged, s));
    J.staged_pending = true;
    uint8_t* d_out = frames;
    const size_t out_bytes = (size_t)n_frames * rows * cols * 3;
    if (!frames_on_device) {
        if ((rc = ensure(h, J.out, out_bytes)) != LF_OK) return rc;
        d_out = static_cast<uint8_t*>(J.out.p);
    }
    {
        StageTimer t(h, ST_JPEG);
        launch_jpeg_decode(g, n_frames, max_blocks, reinterpret_cast<const lf::jpeg::FrameHeader*>(d_stage),
                           reinterpret_cast<const uint32_t*>(d_stage + off_ent), reinterpret_cast<const uint32_t*>(d_stage + off_blk),
                           static_cast<uint8_t*>(J.planes.p), d_out, s);
    }
    LF_HIP_CHECK(h, hipGetLastError());
    if (!frames_on_device) {
        LF_HIP_CHECK(h, hipMemcpyAsync(frames, d_out, out_bytes, hipMemcpyDeviceToHost, s));
        LF_HIP_CHECK(h, hipStreamSynchronize(s));
        J.staged_pending = false;
    }
    if (trace)
        fprintf(stderr, "lf_jpeg_decode_batch: %d frames, %d threads: decode %.2f ms, wait %.2f, pack %.2f (%.1f MB), total host %.2f ms\n",
                n_frames, nt, t_decode, t_wait, t_pack, stage_bytes / 1e6, ms_since(t_begin));
    if (n_failed && !frame_status) {
        lf_set_error(h, first_error, "%d of %d JPEG streams could not be decoded (first status %d)", n_failed, n_frames, first_error);
        return LF_ERR_DECODE;
    }
    return LF_OK;
}

// ---------------------------------------------------------------------------------------- SegmentList glue
extern "C" int lf_serialize_segments(lf_handle* h, const lf_segments* segs, int segs_on_device, int n_frames, int stage,
                                     uint8_t* out, size_t out_capacity, int out_on_device, int64_t* frame_byte_offset)
{
    if (!h) return LF_ERR_NOT_INITIALISED;
    if (!segs || !out || !frame_byte_offset || n_frames < 1 || stage < LF_MSG_DETECTOR || stage > LF_MSG_FILTERED || !segs->frame_offset || !segs->color) {
        lf_set_error(h, LF_ERR_BAD_ARG, "lf_serialize_segments: null argument, n_frames < 1 or unknown stage");
        return LF_ERR_BAD_ARG;
    }
    if (stage == LF_MSG_DETECTOR ? (!segs->pixels_normalized || !segs->normals) : (!segs->ground || (stage == LF_MSG_FILTERED && !segs->keep))) {
        lf_set_error(h, LF_ERR_BAD_ARG, "lf_serialize_segments: the arrays of stage %d are missing", stage);
        return LF_ERR_BAD_ARG;
    }
    LF_HIP_CHECK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const int* d_fo = segs->frame_offset;
    const uint8_t* d_color = segs->color;
    const float* d_pn = segs->pixels_normalized;
    const float* d_nm = segs->normals;
    const double* d_gr = segs->ground;
    const uint8_t* d_keep = segs->keep;
    int rc;
    if (!segs_on_device) {
        const int total = segs->frame_offset[n_frames];
        if (total < 0) { lf_set_error(h, LF_ERR_BAD_ARG, "negative segment count"); return LF_ERR_BAD_ARG; }
        const size_t n = (size_t)total;
        if ((rc = ensure(h, h->m_fo, (size_t)(n_frames + 1) * sizeof(int))) != LF_OK) return rc;
        if ((rc = ensure(h, h->m_color, n + 1)) != LF_OK) return rc;
        LF_HIP_CHECK(h, hipMemcpyAsync(h->m_fo.p, segs->frame_offset, (size_t)(n_frames + 1) * sizeof(int), hipMemcpyHostToDevice, s));
        LF_HIP_CHECK(h, hipMemcpyAsync(h->m_color.p, segs->color, n, hipMemcpyHostToDevice, s));
        d_fo = static_cast<const int*>(h->m_fo.p);
        d_color = static_cast<const uint8_t*>(h->m_color.p);
        if (stage == LF_MSG_DETECTOR) {
            if ((rc = ensure(h, h->m_pn, n * 16 + 16)) != LF_OK) return rc;
            if ((rc = ensure(h, h->m_nm, n * 8 + 8)) != LF_OK) return rc;
            LF_HIP_CHECK(h, hipMemcpyAsync(h->m_pn.p, segs->pixels_normalized, n * 16, hipMemcpyHostToDevice, s));
            LF_HIP_CHECK(h, hipMemcpyAsync(h->m_nm.p, segs->normals, n * 8, hipMemcpyHostToDevice, s));
            d_pn = static_cast<const float*>(h->m_pn.p);
            d_nm = static_cast<const float*>(h->m_nm.p);
        } else {
            if ((rc = ensure(h, h->m_gr, n * 32 + 32)) != LF_OK) return rc;
            LF_HIP_CHECK(h, hipMemcpyAsync(h->m_gr.p, segs->ground, n * 32, hipMemcpyHostToDevice, s));
            d_gr = static_cast<const double*>(h->m_gr.p);
            if (stage == LF_MSG_FILTERED) {
                if ((rc = ensure(h, h->m_keep, n + 1)) != LF_OK) return rc;
                LF_HIP_CHECK(h, hipMemcpyAsync(h->m_keep.p, segs->keep, n, hipMemcpyHostToDevice, s));
                d_keep = static_cast<const uint8_t*>(h->m_keep.p);
            }
        }
    }
    if ((rc = ensure(h, h->m_counts, (size_t)n_frames * sizeof(int))) != LF_OK) return rc;
    if ((rc = ensure(h, h->m_boff, (size_t)(n_frames + 1) * sizeof(long long))) != LF_OK) return rc;
    launch_msg_layout(n_frames, stage, d_fo, d_keep, static_cast<int*>(h->m_counts.p), static_cast<long long*>(h->m_boff.p), s);
    static_assert(sizeof(long long) == sizeof(int64_t), "byte offsets are int64");
    LF_HIP_CHECK(h, hipMemcpyAsync(frame_byte_offset, h->m_boff.p, (size_t)(n_frames + 1) * sizeof(long long), hipMemcpyDeviceToHost, s));
    LF_HIP_CHECK(h, hipStreamSynchronize(s));
    const size_t need = (size_t)frame_byte_offset[n_frames];
    if (need > out_capacity) {
        lf_set_error(h, LF_ERR_CAPACITY, "lf_serialize_segments: %zu bytes needed, %zu available", need, out_capacity);
        return LF_ERR_CAPACITY;
    }
    uint8_t* d_out = out;
    if (!out_on_device) {
        if ((rc = ensure(h, h->m_body, need + 16)) != LF_OK) return rc;
        d_out = static_cast<uint8_t*>(h->m_body.p);
    }
    launch_msg_write(n_frames, stage, d_fo, d_color, d_pn, d_nm, d_gr, d_keep, static_cast<const int*>(h->m_counts.p),
                     static_cast<const long long*>(h->m_boff.p), d_out, s);
    LF_HIP_CHECK(h, hipGetLastError());
    if (!out_on_device) {
        LF_HIP_CHECK(h, hipMemcpyAsync(out, d_out, need, hipMemcpyDeviceToHost, s));
        LF_HIP_CHECK(h, hipStreamSynchronize(s));
    }
    return LF_OK;
}

extern "C" int lf_deserialize_segments(lf_handle* h, const uint8_t* bodies, int bodies_on_device, const int64_t* frame_byte_offset,
                                       int n_frames, lf_segments* out, int out_on_device, int* n_segments)
{
    if (!h) return LF_ERR_NOT_INITIALISED;
    if (!bodies || !frame_byte_offset || !out || n_frames < 1 || !out->frame_offset) {
        lf_set_error(h, LF_ERR_BAD_ARG, "lf_deserialize_segments: null argument or n_frames < 1");
        return LF_ERR_BAD_ARG;
    }
    for (int f = 0; f < n_frames; ++f)
        if (frame_byte_offset[f + 1] < frame_byte_offset[f] + 4) { lf_set_error(h, LF_ERR_DECODE, "body %d is shorter than its count field", f); return LF_ERR_DECODE; }
    LF_HIP_CHECK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const size_t bytes = (size_t)frame_byte_offset[n_frames];
    const size_t max_segs = bytes / 73 + 1;
    int rc;
    const uint8_t* d_body = bodies;
    if (!bodies_on_device) {
        if ((rc = ensure(h, h->m_body, bytes + 16)) != LF_OK) return rc;
        LF_HIP_CHECK(h, hipMemcpyAsync(h->m_body.p, bodies, bytes, hipMemcpyHostToDevice, s));
        d_body = static_cast<const uint8_t*>(h->m_body.p);
    }
    if ((rc = ensure(h, h->m_boff, (size_t)(n_frames + 1) * sizeof(long long))) != LF_OK) return rc;
    if ((rc = ensure(h, h->m_bad, sizeof(int))) != LF_OK) return rc;
    LF_HIP_CHECK(h, hipMemcpyAsync(h->m_boff.p, frame_byte_offset, (size_t)(n_frames + 1) * sizeof(long long), hipMemcpyHostToDevice, s));
    LF_HIP_CHECK(h, hipMemsetAsync(h->m_bad.p, 0, sizeof(int), s));
    lf_segments dev = *out;
    if (!out_on_device) {
        if ((rc = ensure(h, h->m_fo, (size_t)(n_frames + 1) * sizeof(int))) != LF_OK) return rc;
        dev.frame_offset = static_cast<int32_t*>(h->m_fo.p);
        dev.color = nullptr; dev.pixels_normalized = nullptr; dev.normals = nullptr; dev.ground = nullptr;
        if (out->color) { if ((rc = ensure(h, h->m_color, max_segs)) != LF_OK) return rc; dev.color = static_cast<uint8_t*>(h->m_color.p); }
        if (out->pixels_normalized) { if ((rc = ensure(h, h->m_pn, max_segs * 16)) != LF_OK) return rc; dev.pixels_normalized = static_cast<float*>(h->m_pn.p); }
        if (out->normals) { if ((rc = ensure(h, h->m_nm, max_segs * 8)) != LF_OK) return rc; dev.normals = static_cast<float*>(h->m_nm.p); }
        if (out->ground) { if ((rc = ensure(h, h->m_gr, max_segs * 32)) != LF_OK) return rc; dev.ground = static_cast<double*>(h->m_gr.p); }
    }
    const int cap = out->capacity;
    launch_msg_read(n_frames, cap, d_body, static_cast<const long long*>(h->m_boff.p), dev.frame_offset, static_cast<int*>(h->m_bad.p),
                    dev.color, dev.pixels_normalized, dev.normals, dev.ground, s);
    LF_HIP_CHECK(h, hipGetLastError());
    int bad = 0, total = 0;
    LF_HIP_CHECK(h, hipMemcpyAsync(&bad, h->m_bad.p, sizeof(int), hipMemcpyDeviceToHost, s));
    LF_HIP_CHECK(h, hipMemcpyAsync(&total, dev.frame_offset + n_frames, sizeof(int), hipMemcpyDeviceToHost, s));
    LF_HIP_CHECK(h, hipStreamSynchronize(s));
    if (n_segments) *n_segments = total;
    if (bad) { lf_set_error(h, LF_ERR_DECODE, "a SegmentList body's count does not match its length"); return LF_ERR_DECODE; }
    if (total > cap) { lf_set_error(h, LF_ERR_CAPACITY, "%d segments exceed the output capacity %d", total, cap); return LF_ERR_CAPACITY; }
    if (!out_on_device) {
        const size_t n = (size_t)total;
        LF_HIP_CHECK(h, hipMemcpyAsync(out->frame_offset, dev.frame_offset, (size_t)(n_frames + 1) * sizeof(int), hipMemcpyDeviceToHost, s));
        if (out->color && n) LF_HIP_CHECK(h, hipMemcpyAsync(out->color, dev.color, n, hipMemcpyDeviceToHost, s));
        if (out->pixels_normalized && n) LF_HIP_CHECK(h, hipMemcpyAsync(out->pixels_normalized, dev.pixels_normalized, n * 16, hipMemcpyDeviceToHost, s));
        if (out->normals && n) LF_HIP_CHECK(h, hipMemcpyAsync(out->normals, dev.normals, n * 8, hipMemcpyDeviceToHost, s));
        if (out->ground && n) LF_HIP_CHECK(h, hipMemcpyAsync(out->ground, dev.ground, n * 32, hipMemcpyDeviceToHost, s));
        LF_HIP_CHECK(h, hipStreamSynchronize(s));
    }
    return LF_OK;
}

#include "lanefront_keylines.inc"
#include "lanefront_lsdkl.inc"
#include "lanefront_matcher.inc"
#include "lanefront_jpeg_gpu.inc"
