// Types shared by k_edlines.hip and the host-side sequencing (lanefront_keylines.inc).
#pragma once
#include "common.h"

namespace lf {

struct EdOct {
    int W, H;
    int cap;               // pixelNum / 5: anchors, first / second part arrays (:1448)
    int max_edges;         // cap / 20
    int max_lines;         // line records per frame and octave
    int marks_in_lds;
    const uint16_t* g;     // [B][H*W]
    const uint32_t* dxy;   // [B][H*W]
    uint32_t* anchors;     // [B][cap]
    uint32_t* part;        // [B][cap]        first part of the chain being drawn
    uint32_t* chain;       // [B][2*cap]      edge chains, packed x | y << 16
    uint32_t* sid;         // [B][max_edges + 2]
    uint32_t* gmarks;      // [B][(H*W+31)/32] when the marks do not fit LDS
    const uint32_t* aflags; // [B][2][n_cwords] the anchor candidates as k_ed_grad tested them (scan interval 2), or null: k_ed_detect tests them
    int* counts;           // [B][4]: anchors, edges (-1: the detector gave up), lines, status
    float* l_ep;           // [B][max_lines][4]
    double* l_c;           // [B][max_lines]     lineEquation[2] of the normalised equation
    float* l_dir;          // [B][max_lines]
    int* l_npx;            // [B][max_lines]
    float* l_sal;          // [B][max_lines]
    uint8_t* tl;           // [B][tl_stride] the line records in the order the fitting waves finish them (EdTemp)
    size_t tl_stride;      // bytes per frame: max_lines * 48 (8 + 16 + 5 * 4, rounded up to 8)
    int* ework;            // [B][3][max_edges + 2] per chain: lines kept, lines kept before the last one was begun, first line's index
};

struct EdAll { EdOct o[LF_MAX_OCTAVES]; };

struct EdFitParams { int anchor_threshold, scan, min_line_len; double fit_err; };

struct KlOut {
    float* start_end; float* in_octave; float* angle; int32_t* num_pixels; float* line_length; int32_t* octave;
    int32_t* class_id; float* response; float* size; float* pt; float* salience; int32_t* frame;
};


void launch_kl_mask(int n_frames, const int* fo_src, int* fo_dst, int* totals, int capacity, const uint8_t* masks, int rows, int cols, uint8_t* erased,
                    int* kept_count, const KlOut& src, const KlOut& dst, hipStream_t s);
size_t ed_anchor_words(int W, int H);
void launch_ed_grad(int H, int W, int n_frames, const uint8_t* src, const int* taps5, int grad_threshold, uint8_t* blur,
                    uint32_t* dxy, uint16_t* g, hipStream_t s, uint32_t* anchor_flags = nullptr, int anchor_thr = 0);
void ed_resize_tables(int H, int W, int DH, int DW, double scale, int* tab);        // 4 ints per destination column, then 4 per destination row
void launch_ed_resize(int H, int W, int DH, int DW, const int* tab, int n_frames, const uint8_t* src, uint8_t* dst, hipStream_t s);
void launch_pyrdown(int H, int W, int n_frames, const uint8_t* src, uint8_t* dst, hipStream_t s);
void launch_ed_blur_any(int H, int W, int n_frames, const uint8_t* src, const int* taps, int ksize, int* tmp, uint8_t* dst, hipStream_t s);
size_t ed_detect_lds_bytes(int W, int H, int scan, bool* marks_in_lds);
int launch_ed_detect(const EdAll& all, const EdFitParams& fp, int n_octaves, int n_frames, size_t lds_bytes, hipStream_t s);
void launch_ed_slots(const EdAll& all, int n_frames, const uint32_t* maskbits, int Ww, int cap_lines, float* slot_lines, int* counts, int* failed, hipStream_t s);
void launch_kl_count(const EdAll& all, int n_octaves, int n_frames, int* frame_count, int* status, hipStream_t s);
void launch_kl_offsets(int n_frames, const int* frame_count, int capacity, int* frame_offset, int* totals, hipStream_t s);
void launch_kl_assemble(const EdAll& all, int n_octaves, int n_frames, const int* frame_offset, int capacity, const KlOut& out, uint8_t* big, int big_stride,
                        int lds_lines, hipStream_t s);

}  // namespace lf
