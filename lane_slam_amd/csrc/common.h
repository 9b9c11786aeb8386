// lanefront internal declarations shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/lanefront.h"
#include "detmath.h"

namespace lf {

constexpr int kMaxKsize = 9;        // dilation structuring element
constexpr int kMaxGaussTaps = 15;   // LSD Gaussian
constexpr float kNotDef = -1024.0f; // LSD NOTDEF marker (angle plane, degrees)
#ifndef LF_LABEL_ITEMS
#define LF_LABEL_ITEMS 8192
#endif
constexpr int kLabelItems = LF_LABEL_ITEMS;   // problems up to AT LEAST this many defined pixels are split into connected components (LsdParams::label_items:
                                    // a third of the LSD image, see lanefront_api.hip)
#ifndef LF_LABEL_LDS
#define LF_LABEL_LDS 6144
#endif
constexpr int kLabelLds = LF_LABEL_LDS;       // ... in LDS up to this many (LsdParams::label_lds), in the region scratch beyond
constexpr int kCompCap = 1024;      // component list entries per problem (more eligible components: one component)

enum Stage {
    ST_PRE = 0, ST_CANNY, ST_HYST, ST_LSD_GRAD, ST_LSD_ORDER, ST_LSD_GROW, ST_SEGMENTS,
    ST_LBD_GRAD, ST_LBD, ST_ASSOC_PACK, ST_ASSOC, ST_MISC, ST_JPEG, ST_LSD_LABEL, ST_COUNT
};
static_assert(ST_COUNT == LF_N_STAGES, "stage table out of sync with lanefront.h");

struct PreParams {
    int in_rows, in_cols, img_rows, img_cols, top_cutoff, Hc, W;
    int resize;
    int identity_ai;    // ai_scale == 1 and ai_shift == 0 for all channels
    double ifx, ify;
    float ai_scale[3], ai_shift[3];
    int lo[4][3], hi[4][3];
    int ksize, r;
    int j1[kMaxKsize], j2[kMaxKsize];
};

struct CannyParams {
    int Hc, W, Ww;      // Ww = 32-bit words per row of the bit planes
    int low, high;
};

struct LsdParams {
    int Hc, W;          // working image
    int Hs, Ws;         // scaled image
    int Ww;             // words per row of the edge bit plane
    int scaled;         // lsd_scale != 1
    int ntaps, half;    // Gaussian
    double k[kMaxGaussTaps];
    double rho;         // gradient threshold quant / sin(prec)
    double prec, p;     // angle tolerance (rad) and its probability
    double log_nt, log_eps, density_th, scale;
    int min_reg_size, n_bins, refine;
    int cap_lines;
    int label_items;    // k_lsd_label's capacity: problems with more defined pixels are grown as one component (labels are u16; sizes the region scratch)
    int label_items_max; // = label_items (rounds 2 - 3: a handle moved label_items up to this)
    int label_lds;      // problems of up to this many defined pixels are labelled in LDS, the others in the region scratch (k_lsd_label)
    int rec_cap;        // entries per problem of every per-problem list (records, compact arrays, seed lists, sort scratch): the stride of those
                        // arrays.  Hs * Ws holds any problem; a batch handle starts lower and grows when a batch needs more (lanefront_api.hip)
    // Round 6: what a region that STARTS at a pixel begins its two float sums with -- (float) cos / sin of the pixel's angle as a double
    // (region_grow adds every other pixel with the cosine of the angle rounded to float: the c_cs / c_sn pairs).  k_lsd_grad works it out
    // per record beside those, k_lsd_order gathers it ([rec_cap] float pairs per problem, like c_cs), k_lsd_grow loads the seed's pair
    // instead of evaluating a double sine and cosine at every region start.  Null: k_lsd_grow evaluates them (the octave LSD path).
    float* r_sd = nullptr;
    float* c_sd = nullptr;
};

struct SegParams {
    int Hc, W, img_rows, img_cols, top_cutoff, cap_lines;
    double rx, ry, cut;
    double cw, ch;
    double H[9], K[9], D[5], RR[9];
    double lanewidth, linewidth_white, linewidth_yellow, d_min, d_max, phi_min, phi_max;
};

// device resize tables for the LSD bilinear step (cv::resize INTER_LINEAR on CV_64F)
struct ResizeTables {
    const int* xofs;      // [Ws]
    const float* xa;      // [2*Ws]
    const int* y0;        // [Hs]
    const int* y1;        // [Hs]
    const float* yb;      // [2*Hs]
    int xmax;
};

}  // namespace lf

#ifdef __HIPCC__
// XCD-aware tile order for the streaming stencil kernels.  Workgroups are dealt to the 8 XCDs round robin by their
// linear id, and every XCD has its own L2: in the natural order the tiles of one tile ROW -- which share the 128-byte
// lines their left / right halos straddle (a 64-pixel row segment of a 1 byte/pixel plane is half a line) -- sit on
// different XCDs and each L2 fetches its own copy.  Here the 8 XCDs take the 8 tile rows of a group of 8 * nx
// consecutive ids, one whole row each: lines are shared inside an L2, while the chip as a whole still sweeps the frames
// in order (handing each XCD a contiguous eighth of ALL tiles made the fetched bytes equal the algorithmic ones but
// cost 3-10 % of kernel time).
__device__ __forceinline__ void lf_xcd_tile(int& bx, int& by, int& bz)
{
    const unsigned nx = gridDim.x, ny = gridDim.y, total = nx * ny * gridDim.z;
    unsigned b = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
    const unsigned group = 8u * nx;
    if (b < total / group * group) {
        const unsigned g = b / group, r = b - g * group;
        b = g * group + (r & 7u) * nx + (r >> 3);
    }
    bx = (int)(b % nx);
    const unsigned q = b / nx;
    by = (int)(q % ny);
    bz = (int)(q / ny);
}
#endif

#define LF_HIP_CHECK(h, expr)                                                              \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            lf_set_error((h), LF_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                         __FILE__, __LINE__);                                              \
            return LF_ERR_HIP;                                                             \
        }                                                                                  \
    } while (0)

extern "C" void lf_set_error(lf_handle* h, int code, const char* fmt, ...);

// kernel launchers (one per translation unit)
namespace lf {
void launch_pre_gray(const PreParams& p, const uint8_t* frames, int n_frames, uint8_t* gray, hipStream_t s);
void launch_pre(const PreParams& p, const uint8_t* frames, int n_frames, uint32_t* bgr, uint8_t* gray,
                uint32_t* maskbits, const int* sdiv, const int* hdiv, hipStream_t s);
void launch_canny(const CannyParams& p, const uint32_t* bgr, int n_frames, uint32_t* strong, uint32_t* weak,
                  hipStream_t s);
int launch_hysteresis(const CannyParams& p, int n_frames, uint32_t* strong, const uint32_t* weak, hipStream_t s);
void launch_bgrx_to_bgr(int n_pix, const uint32_t* bgrx, uint8_t* bgr, hipStream_t s);
void launch_edges_u8(const CannyParams& p, int n_frames, const uint32_t* bits, uint8_t* edges, hipStream_t s);
void launch_lsd_grad(const LsdParams& p, const ResizeTables& rt, int n_frames, const uint32_t* edge_bits,
                     const uint32_t* mask_bits, uint32_t* r_addr, float* r_deg, double* r_mod, double* r_cs, double* r_sn,
                     int* n_rec, unsigned long long* maxgrad, int max_nsx, int max_nsy, uint32_t* list, int* list_count,
                     uint32_t* l_addr, double* l_mod, int* n_low, int* rec_need, bool counters_zeroed, hipStream_t s);
void launch_lsd_grad_gray(const LsdParams& p, const ResizeTables& rt, int n_frames, const uint8_t* gray, uint32_t* r_addr, float* r_deg,
                          double* r_mod, double* r_cs, double* r_sn, int* n_rec, unsigned long long* maxgrad, int max_nsx, int max_nsy,
                          uint32_t* list, int* list_count, uint32_t* l_addr, double* l_mod, int* n_low, hipStream_t s);
// the seed order of OpenCV >= 3.2 (k_lsd_seed32.hip): rewrites order_a after launch_lsd_order; l_*: k_lsd_grad's "low" records
bool lsd_seed32_supported(const LsdParams& p);
void launch_lsd_seed32(const LsdParams& p, int n_frames, int* n_rec, int* norder, int* rec_need, const unsigned long long* maxgrad, const uint32_t* c_xy,
                       const double* c_mod, const uint32_t* l_addr, double* l_mod, const int* n_low,
                       unsigned long long* sort_a, unsigned long long* sort_b, uint32_t* order_a, uint32_t* order_b, int big, hipStream_t s);
size_t std_sort_debug_words(int n);
void launch_std_sort_debug(const uint32_t* E, uint32_t* work, int n, int* count, hipStream_t s);
void launch_lsd_order(const LsdParams& p, int n_frames, const uint32_t* r_addr, const float* r_deg, const double* r_mod,
                      const double* r_cs, const double* r_sn, int* n_rec, const unsigned long long* maxgrad,
                      unsigned long long* sort_a, unsigned long long* sort_b, uint32_t* order_a, uint32_t* order_b,
                      int* norder, uint32_t* c_xy, float* c_deg, double* c_mod, double* c_cs, double* c_sn,
                      int* row_start, hipStream_t s);
void launch_lsd_dense_debug(const LsdParams& p, int n_frames, const int* norder, const uint32_t* c_xy, const float* c_deg,
                            const double* c_mod, float* ang, double* mod, hipStream_t s);
void launch_lsd_rank(int n_prob, const int* norder, int* perm, hipStream_t s);
void launch_lsd_label(const LsdParams& p, int n_frames, const int* norder, const uint32_t* c_xy, const int* row_start,
                      uint16_t* c_label, uint16_t* comp_list, int* comp_count, int* comp_key, uint32_t* scratch, hipStream_t s);
size_t lsd_grow_reg_stride(const LsdParams& p);
void launch_lsd_grow(const LsdParams& p, int n_frames, const uint32_t* order, const int* norder, const uint32_t* c_xy,
                     const float* c_deg, const double* c_mod, const double* c_cs, const double* c_sn,
                     const int* row_start, const uint16_t* c_label, const uint16_t* comp_list, const int* comp_count, int comp_cap,
                     uint32_t* reg, uint32_t* gused, float* tmp_lines, int* tmp_tags, float* lines, int* counts, const int* perm,
                     double* pend_rec, int* pend_tag, int* pend_count, int lds_kb, bool mixed, int bitmap, hipStream_t s);   // bitmap: 0 = row-list form, 1 = bit-plane form, > 1 = bit-plane form with that many USED bits (tests)
int lsd_grow_def_lds(const LsdParams& p, int lds_kb);   // defined pixels of a problem that fit k_lsd_grow's LDS slice of lds_kb KB
constexpr int kGrowLdsKb[3] = { 13, 28, 40 };           // the slice sizes a handle moves between (k_lsd_grow.hip)
int lsd_grow_pend_cap(const LsdParams& p);      // entries per problem of the pending-region list (k_lsd_eval)
void launch_seg_offsets(int n_frames, int cap_lines, const int* counts, int* seg_offset, int* frame_offset,
                        int* overflow, const int* norder, int cap_small, int cap_medium, hipStream_t s);
void launch_segments(const SegParams& p, int n_frames, const float* slot_lines, const int* counts,
                     const int* seg_offset, const uint32_t* maskbits, int Ww, lf_segments out, int* seg_frame,
                     double* normals64, float* centers, hipStream_t s);
void launch_lbd_grad(int Hc, int W, int n_frames, const uint8_t* gray, uint32_t* dxy, hipStream_t s);
// gradient planes of the octaves of a KeyLine batch: [B][H*W] dx | dy << 16 each
struct LbdPlanes { const uint32_t* base[LF_MAX_OCTAVES]; int W[LF_MAX_OCTAVES], H[LF_MAX_OCTAVES]; };
void launch_lbd_keylines(const LbdPlanes& planes, int n_cap, int n_frames, const int* n_lines, const float* in_octave4, const float* angle, const int* npx,
                         const int* octave, const int* frame, const float* gauss_g, const float* gauss_l, float* desc, uint8_t* code,
                         hipStream_t s, int wband = 7);
void launch_lbd_split_debug(size_t n, const uint32_t* dxy, int16_t* dx, int16_t* dy, hipStream_t s);
void launch_lbd(int Hc, int W, int n_seg_cap, const int* n_seg, const float* lines, const int* seg_frame,
                const uint32_t* dxy, const float* gauss_g, const float* gauss_l,
                float* desc, uint8_t* code, hipStream_t s, int wband = 7);
int lbd_max_width_of_band();
// ---- associator (k_assoc.hip): packed operands = [rows][256] int8 code bytes + [rows][32] int8 ninth-step operand
size_t assoc_rows_padded_m(int nm);
// Packed map operand layout (k_assoc.hip): blocked by the associator's 64-row LDS tile, [tile][16-byte chunk][row][16 B];
// byte `byte` (0..255) of map row `row` lives at
__host__ __device__ inline size_t assoc_map_offset(size_t row, int byte)
{
    return (row >> 6) * 16384 + (size_t)(byte >> 4) * 1024 + (row & 63) * 16 + (size_t)(byte & 15);
}
// The ungated associator runs on the FP4 matrix instruction: one e2m1 nibble per code bit, 128 bytes per map row, blocked the
// same way with 8 KB tiles: byte `byte` (0..127) of map row `row` lives at
__host__ __device__ inline size_t assoc_map_offset_fp4(size_t row, int byte)
{
    return (row >> 6) * 8192 + (size_t)(byte >> 4) * 1024 + (row & 63) * 16 + (size_t)(byte & 15);
}
// eight code bits -> eight e2m1 nibbles: bit 0 -> +1.0 (0x2), bit 1 -> -1.0 (0xA); bit j in nibble j
__host__ __device__ inline uint32_t assoc_fp4_expand(uint32_t byte)
{
    uint32_t t = (byte | (byte << 12)) & 0x000f000fu;
    t = (t | (t << 6)) & 0x03030303u;
    t = (t | (t << 3)) & 0x11111111u;
    return (t << 3) | 0x22222222u;
}
// per-caller scratch of the associator (k_assoc.hip): one key per query and map chunk + arrival counters; sized by
// launch_assoc_core itself
struct AssocScratch {
    unsigned int* part = nullptr; int* done = nullptr; size_t cap_part = 0, cap_blocks = 0;
    // the last launch_assoc_core's split of the map (the tie pass reads part[] with it) and the tie pass's query lists
    int qblocks = 0, splits = 0, m_chunk = 0;
    int* tie_list = nullptr; size_t cap_list = 0;
    unsigned long long* tie_res = nullptr;        // set by the caller before launch_assoc_core when launch_assoc_ties follows: the merge step then writes the tie pass's query lists
};
void assoc_scratch_free(AssocScratch& w);
void launch_assoc_pack_map(const uint8_t* codes, const uint8_t* colors, int n, int n_pad, int fp4, int8_t* x, int8_t* cx, hipStream_t s);
hipError_t launch_assoc_core(const uint8_t* q, const uint8_t* qcolor, int nq, const int8_t* mx, const int8_t* mcx, int nm,
                             const int* nm_dev, int gating, int max_distance, AssocScratch& w, int32_t* idx, float* dist, hipStream_t s);
// the reference's tie rule as a second pass over the packed map (k_assoc_ties.hip); after launch_assoc_core on the same stream
hipError_t launch_assoc_ties(const uint8_t* q, const uint8_t* qcolor, int nq, const int8_t* mx, const uint8_t* mcode, const uint8_t* mcolor,
                             int nm, const int* nm_dev, int gating, AssocScratch& w, unsigned long long* res, int32_t* idx, const float* dist, hipStream_t s);
hipError_t launch_assoc(const uint8_t* q, int nq, const uint8_t* m, int nm, int8_t* mx, int8_t* mcx, AssocScratch& w, int32_t* idx,
                        float* dist, hipStream_t s);
void launch_assoc_nomatch(int nq, int32_t* idx, float* dist, hipStream_t s);
// ---- anti-instagram colour clustering (k_kmeans.hip)
void launch_kmeans(const uint8_t* bgr, int n, int k, const double* init, int max_iter, double tol_rel, uint8_t* lab, double* out,
                   long long* counts, int* status, hipStream_t s);
// ---- live map (k_map.hip)
struct MapDevice {
    int capacity, policy, kept_only, merge_distance, when_full;
    int fp4;                       // packed operands are e2m1 nibbles (ungated maps) instead of int8 bytes + ninth-step rows
    uint8_t* code; uint8_t* color; double* ground; int* hits; int* last_seen; int* winner;
    int8_t* mx; int8_t* mcx;
    int* state;                    // 16 ints: [0] size [1] head [2] flags of the latest failing update [3] n_app [4] n_ref [5] old head
                                   // [6] old size [7] step [8] failing updates so far [9], [10] accumulators (k_map.hip)
    unsigned long long* totals;    // [0] appended [1] refreshed
};
void launch_fill_i32(int* p, size_t n, int v, hipStream_t s);
void launch_map_pack_block(int n, int n_frames, const int* frame_offset, const uint8_t* code, const uint8_t* color,
                           const uint8_t* keep, const double* ground, const int32_t* idx, const float* dist,
                           const double* pose4, int step, uint8_t* block, hipStream_t s);
void launch_map_overflow_block(int n, int n_frames, int step, uint8_t* block, hipStream_t s);
void launch_map_seed_block(int n, const uint8_t* code, const uint8_t* color, const double* ground, uint8_t* block, hipStream_t s);
void launch_map_update(const MapDevice& md, const uint8_t* blocks, int n_blocks, int block_rows, int force_append, int* act,
                       hipStream_t s);
// knnMatch / radiusMatch forms (k_knn.hip)
void launch_select_queries(const uint8_t* q, const uint8_t* mask, int nq, uint8_t* out, int32_t* qidx, int* n_out, hipStream_t s);
void launch_knn(const uint8_t* q, int nq, const uint8_t* m, int nm, int k, int max_distance, int mih, int32_t* idx, float* dist, hipStream_t s);
void launch_radius(const uint8_t* q, int nq, const uint8_t* m, int nm, int max_distance, int32_t* hist, int32_t* count, int32_t* offsets,
                   int* total, int cap, int mih, int32_t* idx, float* dist, hipStream_t s);
void launch_assoc_float(const float* q, int nq, const float* m, int nm, float* qn, float* mn,
                        unsigned long long* best, int32_t* idx, float* dist, hipStream_t s);
// ---- JPEG ingest (k_jpeg.hip)
namespace jpeg { struct FrameHeader; }
struct JpegGeom { int rows, cols, Wp, Hp; int first_row = 0; };     // Wp x Hp: padded component plane (multiples of 16); first_row: rows above it are not wanted (the dense IDCT and the colour kernel skip them)
void launch_jpeg_decode(const JpegGeom& g, int n_frames, int max_blocks, const jpeg::FrameHeader* hdrs,
                        const uint32_t* entries, const uint32_t* block_end, uint8_t* planes, uint8_t* frames,
                        hipStream_t s);

void launch_jpeg_color(const JpegGeom& g, int n_frames, const void* hdrs, size_t hdr_stride, const uint8_t* planes, uint8_t* frames, hipStream_t s);
// entropy decoding on the device (k_jhuff.hip)
namespace jpeg { struct DevFrame; }
void launch_jh_decode(const JpegGeom& g, int n_frames, int max_blocks, size_t max_scan_len, jpeg::DevFrame* frames, const uint8_t* bytes, uint8_t* clean,
                      uint32_t* seg_begin, void* info, uint32_t* sub, int16_t* coef, int* status, uint8_t* planes, hipStream_t s);
size_t jh_info_bytes();

// ---- SegmentList wire bodies (k_msgs.hip)
void launch_msg_layout(int n_frames, int stage, const int* frame_offset, const uint8_t* keep, int* counts,
                       long long* byte_offset, hipStream_t s);
void launch_msg_write(int n_frames, int stage, const int* frame_offset, const uint8_t* color, const float* pixels_normalized,
                      const float* normals, const double* ground, const uint8_t* keep, const int* counts,
                      const long long* byte_offset, uint8_t* out, hipStream_t s);
void launch_msg_read(int n_frames, int capacity, const uint8_t* body, const long long* byte_offset, int* frame_offset,
                     int* bad, uint8_t* color, float* pixels_normalized, float* normals, double* ground, hipStream_t s);

}  // namespace lf
