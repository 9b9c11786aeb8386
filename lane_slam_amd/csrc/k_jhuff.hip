// JPEG entropy decoding ON THE DEVICE: the Huffman-coded scan of every frame of a batch is
// decoded by the GPU, the host only parses the headers (jpeg_entropy.cpp: prepare_device_frame).  Replaces the decode
// half of duckietown_utils.jpg.image_cv_from_jpg = cv2.imdecode (ref: src/duckietown/include/duckietown_utils/jpg.py:21-31,
// called per frame from src/line_detector/src/line_detector_node.py:153-158), i.e. libjpeg-turbo's baseline decoder
// (ITU-T T.81 F.2.2); same results, bit for bit, as the host decoder and oracle/lf_oracle_jpeg.c.
//
// A Huffman stream has no random access: where a code starts is only known once everything before it is decoded.  But
// it SELF-SYNCHRONISES: a decoder started at a wrong position falls into step with the true sequence of codes after a
// few symbols.  So the scan is cut into subsequences of 128 bytes, one thread each:
//   k_jh_unstuff   per frame: removes the stuffed zeros (FF 00 -> FF) and the restart markers, notes where every restart
//                  interval begins (an interval starts byte aligned, at an MCU boundary, with a known decoder state) and
//                  where the scan ends -- three passes over ~50 KB, prefix counts by ballot / LDS
//   k_jh_decode    per frame, 256 threads.  Pass 0: every thread decodes its subsequence from an ASSUMED state (first
//                  bit, start of an MCU) and records the state in which it crosses into the next subsequence.  Passes
//                  1, 2, ...: a thread whose predecessor handed over a different state than the one it started from decodes
//                  again from that state; when nothing changes any more every subsequence has its true entry state
//                  (subsequence 0 of an interval is exact from the start, so after pass p the first p + 1 are: at most as
//                  many passes as subsequences; measured on 640x480 camera-like frames: 14 - 25, because the block's place in the MCU, part of the
//                  state, never re-synchronises by itself -- DESIGN.md section 8).  Then the blocks completed per subsequence are
//                  prefix-summed and a last decode writes the quantised coefficients (int16, natural order, dense 64 per
//                  block; DC still as differences) and performs the checks of a sequential decoder: invalid code, run past
//                  63, DC size > 11, reading beyond the interval, a restart interval of the wrong length.
//   k_jh_dc        DC prediction: running sum per component over the blocks of a restart interval
//   k_jpeg_idct_dense  dequantisation + the 13-bit integer inverse DCT of k_jpeg.hip on the dense blocks
// Chroma upsampling and colour conversion are k_jpeg.hip's k_jpeg_color, unchanged.
#include <cstdio>
#include <cstdlib>
#include "common.h"
#include "jpeg_entropy.h"

namespace lf {

constexpr int JH_T = 256;          // k_jh_unstuff
#ifndef LF_JH_THREADS
#define LF_JH_THREADS 512
#endif
// k_jh_decode's workgroup.  512 since round 6: alone the kernel is 12 % slower than with 1 024 (decode 142 k -> 126 k frames/s), behind the front end
// it is what a CU can take in beside the other batches' kernels -- decode + front end 88.5 k -> 93.3 k frames/s, same call (768: 89.4 k, 384: 92.5 k,
// 256: 85 k).  Until round 5's Jacobi form 1 024 was the fastest both ways.
constexpr int JH_TD = LF_JH_THREADS;
constexpr int JH_LDS_CLEAN = 96 * 1024;     // scans up to this many clean bytes are decoded out of LDS
#ifndef LF_JH_SB
#define LF_JH_SB 48
#endif
constexpr int JH_SB = LF_JH_SB;    // clean bytes per subsequence

struct JhInfo { int clean_len, n_seg, n_sub, err; };

__constant__ uint8_t c_zigzag[64] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

// ---------------------------------------------------------------------------------------------- k_jh_unstuff
// classification of raw byte i: 0 dropped (stuffed zero / marker byte), 1 kept, 2 first byte of an RSTn marker, 3 terminal
// PAD: the raw scan staged in LDS with one spare dword after every 32 (byte b at b + 4 * (b / 128)): the lanes of a wave
// walk chunks 128 bytes apart, without the padding all in one bank
template <bool PAD>
__device__ __forceinline__ uint32_t jh_raw(const uint8_t* raw, uint32_t i) { return raw[PAD ? i + ((i >> 7) << 2) : i]; }

template <bool PAD>
__device__ __forceinline__ int jh_classify(const uint8_t* raw, uint32_t i, uint32_t L, int restart)
{
    const uint32_t cur = jh_raw<PAD>(raw, i);
    if (cur == 0xFFu) {
        const uint32_t nxt = i + 1 < L ? jh_raw<PAD>(raw, i + 1) : 0xD9u;
        if (nxt == 0u) return 1;
        if (restart && nxt >= 0xD0u && nxt <= 0xD7u) return 2;
        return 3;
    }
    if (i > 0 && (cur == 0u || (restart && cur >= 0xD0u && cur <= 0xD7u)) && jh_raw<PAD>(raw, i - 1) == 0xFFu) return 0;
    return 1;
}

extern __shared__ uint32_t jh_dyn[];                         // k_jh_unstuff: the raw scan; k_jh_decode: the clean scan (when they fit)

template <int NT> struct JhUShared { uint32_t s_term; int s_wk[NT / 64], s_wm[NT / 64], s_ck, s_cm, s_err; };

// NT threads: k_jh_unstuff's 256, or the decoder's workgroup (k_jh_decode unstuffs its own frame when the raw scan fits LDS: round 6)
template <bool PAD, int NT>
__device__ __forceinline__ void jh_unstuff_body(JhUShared<NT>& sh, const jpeg::DevFrame& F, JhInfo* I, const uint8_t* raw, uint8_t* __restrict__ out,
                                                uint32_t* __restrict__ sb)
{
    uint32_t& s_term = sh.s_term; int* s_wk = sh.s_wk; int* s_wm = sh.s_wm; int& s_ck = sh.s_ck; int& s_cm = sh.s_cm; int& s_err = sh.s_err;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t L = F.scan_len;
    if (t == 0) { s_term = L; s_ck = 0; s_cm = 0; s_err = 0; }
    __syncthreads();
    // the scan ends at the first marker that is not a restart marker (EOI, normally)
    uint32_t term = L;
    for (uint32_t i = t; i < L; i += NT)
        if (jh_raw<PAD>(raw, i) == 0xFFu && jh_classify<PAD>(raw, i, L, F.restart) == 3) { term = i; break; }
    atomicMin(&s_term, term);
    __syncthreads();
    const uint32_t T = s_term;
    const int expected_seg = F.restart > 0 ? (F.n_mcu + F.restart - 1) / F.restart : 1;
    // chunks of 128 raw bytes per thread, rounds of 256 chunks: kept bytes and restart markers before every chunk
    for (uint32_t base = 0; base < T; base += (uint32_t)NT * 128u) {
        const uint32_t c0 = base + (uint32_t)t * 128u;
        const uint32_t c1 = c0 + 128u < T ? c0 + 128u : T;
        int kept = 0, marks = 0;
        for (uint32_t i = c0; i < c1; ++i) { const int cl = jh_classify<PAD>(raw, i, L, F.restart); kept += cl == 1; marks += cl == 2; }
        int ik = kept, im = marks;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int a = __shfl_up(ik, d), b = __shfl_up(im, d); if (lane >= d) { ik += a; im += b; } }
        if (lane == 63) { s_wk[wave] = ik; s_wm[wave] = im; }
        __syncthreads();
        int pos = s_ck, ord = s_cm;
        for (int w = 0; w < wave; ++w) { pos += s_wk[w]; ord += s_wm[w]; }
        pos += ik - kept; ord += im - marks;
        for (uint32_t i = c0; i < c1; ++i) {
            const int cl = jh_classify<PAD>(raw, i, L, F.restart);
            if (cl == 1) out[pos++] = (uint8_t)jh_raw<PAD>(raw, i);
            else if (cl == 2) {
                // RSTm: m counts modulo 8 (T.81 E.1.4); the next interval starts at the next clean byte
                if ((int)jh_raw<PAD>(raw, i + 1) != 0xD0 + (ord & 7)) atomicOr(&s_err, 1);
                if (ord + 1 < F.max_seg) sb[ord + 1] = (uint32_t)pos;
                ++ord;
            }
        }
        __syncthreads();
        if (t == NT - 1) { s_ck = pos; s_cm = ord; }
        __syncthreads();
    }
    const int clean_len = s_ck, found = s_cm + 1;
    for (int i = t; i < 32; i += NT) out[clean_len + i] = 0;          // the bit reader may look (never consume) past the end
    if (t == 0) {
        int err = s_err;
        int n_seg = found;
        if (found < expected_seg) err |= 1;                               // a restart marker is missing
        if (n_seg > expected_seg) n_seg = expected_seg;                   // whatever follows the last MCU is ignored, as a sequential decoder does
        if (n_seg > F.max_seg - 1) { n_seg = F.max_seg - 1; err |= 1; }
        sb[0] = 0u;
        // the interval after the last one used starts where the data of interest ends
        if (n_seg == found) sb[n_seg] = (uint32_t)clean_len;
        I->clean_len = clean_len; I->n_seg = n_seg; I->n_sub = 0; I->err = err;
    }
}

__global__ __launch_bounds__(JH_T) void k_jh_unstuff(const jpeg::DevFrame* __restrict__ frames, const uint8_t* __restrict__ bytes,
                                                     uint8_t* __restrict__ clean, uint32_t* __restrict__ seg_begin, JhInfo* __restrict__ info,
                                                     int lds_raw_bytes)
{
    __shared__ JhUShared<JH_T> sh;
    const int f = blockIdx.x, t = threadIdx.x;
    const jpeg::DevFrame& F = frames[f];
    JhInfo* I = info + f;
    if (!F.hdr.valid) { if (t == 0) { I->clean_len = 0; I->n_seg = 0; I->n_sub = 0; I->err = 0; } return; }
    const uint8_t* raw = bytes + F.scan_off;
    uint8_t* out = clean + F.clean_off;
    uint32_t* sb = seg_begin + F.seg_off;
    const uint32_t L = F.scan_len;
    if (lds_raw_bytes > 0 && (int)(L + L / 32 + 64) <= lds_raw_bytes) {
        // the raw scan into LDS once (coalesced dwords; scan_off is a multiple of 16): three passes read it byte by byte
        const uint32_t* src = reinterpret_cast<const uint32_t*>(raw);
        const int n4 = (int)((L + 3) / 4);
        for (int i = t; i < n4; i += JH_T) jh_dyn[i + (i >> 5)] = src[i];
        __syncthreads();
        jh_unstuff_body<true, JH_T>(sh, F, I, reinterpret_cast<const uint8_t*>(jh_dyn), out, sb);
    } else {
        jh_unstuff_body<false, JH_T>(sh, F, I, raw, out, sb);
    }
}

// ---------------------------------------------------------------------------------------------- k_jh_decode
struct JhTabs { jpeg::HuffDev t[8]; };

// PAD (jh_span): the stream sits in LDS with one spare dword after every 32 (dword d at d + d / 32): the 64 lanes of a wave read
// from subsequences a fixed distance apart, which without the padding is the SAME bank for many of them
// one Huffman code from the top of w: returns length (0: no code of up to 16 bits matches) and the symbol
__device__ __forceinline__ int jh_code(const jpeg::HuffDev& h, uint32_t w, int& sym)
{
    const uint32_t e = h.fast[w >> 23];
    if (e) { sym = (int)(e & 255u); return (int)(e >> 8); }
    // codes of 10 .. 16 bits: the SHORTEST length whose prefix is a code.  All seven limits are fetched at once (the loop with an early
    // exit was a chain of up to seven dependent LDS trips, and with 64 lanes on different symbols some lane of a wave takes this path
    // at nearly every step)
    const int code16 = (int)(w >> 16);
    int len = 0;
#pragma unroll
    for (int l = 16; l >= 10; --l)
        if ((code16 >> (16 - l)) <= h.maxcode[l]) len = l;
    sym = 0;
    if (len) sym = (int)h.vals[((code16 >> (16 - len)) + h.delta[len]) & 255];
    return len;
}

struct JhState { uint32_t bit; int blk, k; };     // next symbol starts at `bit`; block `blk` of the MCU; coefficient k (0: DC code next)

// Decodes the symbols that START in [st.bit, limit).  WRITE = false: speculative (errors only re-align), returns the number of
// blocks completed.  WRITE = true: true state; coefficients go to coef (block index cur, stops at end_block); err is set on
// anything a sequential decoder rejects.
struct JhSel { int luma, bpm; uint32_t tsel; };   // tsel: DC table of component c in bits 4c .. 4c+3, AC table in bits 12 + 4c ..

template <bool WRITE, bool PAD>
__device__ __forceinline__ int jh_span(const JhTabs& tabs, const JhSel sel, const uint8_t* clean, JhState& st, uint32_t limit,
                                       uint32_t end_bit, int16_t* __restrict__ coef, int cur, int end_block, int& err)
{
    int done = 0;
    uint32_t bit = st.bit;
    int blk = st.blk, k = st.k;
    // The stream goes through two dwords in registers (wcur, wnxt: big-endian order restored) and a bit offset sh into wcur, the dword
    // after them always on its way (pre: raw, its bytes are swapped when it moves up -- swapping at the load made every refill wait
    // for it): a symbol's only dependent trip to LDS is its code-table lookup, and the look at the next 32 bits is one funnel shift
    // (round 5; a 64-bit left-aligned buffer before: two quarter-rate 64-bit shifts per symbol).  nd: the dword `pre` holds (clamped
    // to the interval's last dword + 1: reads never go past what the interval owns plus one dword, as before).
    const uint32_t* p = reinterpret_cast<const uint32_t*>(clean);
    const uint32_t dmax = (end_bit >> 5) + 1u;
    uint32_t nd = bit >> 5;
    auto fetch = [&](uint32_t d) { const uint32_t dc_ = d < dmax ? d : dmax; return p[PAD ? dc_ + (dc_ >> 5) : dc_]; };
    // sh = bits of wcur already consumed, kept in 1 .. 32 (32: wcur is used up and the look is wnxt itself), so that the look is ONE
    // v_alignbit_b32 without a special case: a stream position on a dword boundary starts with wcur = nothing, wnxt = that dword
    uint32_t sh = bit & 31u;
    if (sh == 0u) { sh = 32u; --nd; }
    uint32_t wcur = sh == 32u ? 0u : __builtin_bswap32(fetch(nd)), wnxt = __builtin_bswap32(fetch(nd + 1));
    nd += 2;
    uint32_t pre = fetch(nd);
    // the tables of the current block: DC and AC table of its component (their addresses change with the block, not with the symbol)
    int comp = blk < sel.luma ? 0 : blk - sel.luma + 1;
    const jpeg::HuffDev* hdc = &tabs.t[(sel.tsel >> (4 * comp)) & 15u];
    const jpeg::HuffDev* hac = &tabs.t[(sel.tsel >> (4 * comp + 12)) & 15u];
    while (bit < limit) {
        if (WRITE && cur >= end_block) break;
        // one symbol = one Huffman code + its magnitude bits (at most 16 + 15 bits: one 32-bit look).  DC and AC share the
        // path: DC is "run 0, size = the symbol" (T.81 F.2.2.1), AC "run = high nibble, size = low nibble" with EOB / ZRL
        const bool dc = k == 0;
        const uint32_t w = __builtin_amdgcn_alignbit(wcur, wnxt, 32u - sh);      // the next 32 bits (sh = 32: a shift by 0 = wnxt)
        int sym;
        const int len = jh_code(*(dc ? hdc : hac), w, sym);
        const int run = dc ? 0 : sym >> 4, sz = dc ? sym : sym & 15;
        const int kn = k + run;
        const bool bad = len == 0 || (dc && sym > 11) || (!dc && sz != 0 && kn > 63);
        int n;
        if (bad) {
            if (WRITE) { err |= 2; break; }
            n = 1;                                      // speculative decode out of step: slide on, the true state arrives later
            ++err;                                      // (speculative: err counts these)
        } else {
            if (WRITE && sz) {
                const int v = (int)((w << len) >> (32 - sz));
                coef[(size_t)cur * 64 + c_zigzag[kn]] = (int16_t)(v < (1 << (sz - 1)) ? v - (1 << sz) + 1 : v);
            }
            n = len + sz;
            k = (!dc && sz == 0) ? (run == 15 ? k + 16 : 64) : kn + 1;           // ZRL / EOB / a coefficient
        }
        bit += (uint32_t)n;
        sh += (uint32_t)n;                              // n <= 16 + 15: at most one dword moves up
        if (sh > 32u) {
            sh -= 32u;
            wcur = wnxt;
            wnxt = __builtin_bswap32(pre);
            ++nd;
            pre = fetch(nd);
        }
        if (bad) continue;
        if (WRITE && bit > end_bit) { err |= 4; break; }                      // bits that are not in the interval were consumed
        if (k >= 64) {
            k = 0;
            blk = blk + 1 == sel.bpm ? 0 : blk + 1;
            ++done; ++cur;
            comp = blk < sel.luma ? 0 : blk - sel.luma + 1;
            hdc = &tabs.t[(sel.tsel >> (4 * comp)) & 15u]; hac = &tabs.t[(sel.tsel >> (4 * comp + 12)) & 15u];
        }
    }
    st.bit = bit; st.blk = blk; st.k = k;
    return done;
}

constexpr int JH_DL = 2048;         // subsequences of a frame the dirty list of a correction pass holds (a 96 KB scan: what fits LDS)
struct JhShared { JhTabs tabs; int s_changed, s_redo, s_err, s_nsub, s_wsum[JH_TD / 64], s_carry; int s_dc[JH_TD / 64][4], s_dcc[JH_TD / 64 + 1][3];
                  int s_nd; uint16_t s_dlist[JH_DL]; unsigned long long s_wmap[JH_TD / 64]; };

// LDS = true: the clean scan sits in jh_dyn (the compiler sees an LDS address: ds_read instead of flat loads)
template <bool LDS>
__device__ __forceinline__ void jh_frame(JhShared& sh, jpeg::DevFrame& F, JhInfo* I, const uint8_t* clean_global, const uint32_t* sb,
                                         uint32_t* __restrict__ sub_all, int16_t* __restrict__ coef_all, int* __restrict__ status,
                                         int getenv_debug, int f)
{
    constexpr int JH_T = JH_TD;                              // (this kernel's workgroup size)
    JhTabs& tabs = sh.tabs;
    int& s_changed = sh.s_changed; int& s_redo = sh.s_redo; int& s_err = sh.s_err; int& s_nsub = sh.s_nsub; int& s_carry = sh.s_carry;
    int* s_wsum = sh.s_wsum;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint8_t* clean = LDS ? reinterpret_cast<const uint8_t*>(jh_dyn) : clean_global;
    const int n_seg = I->n_seg;
    JhSel sel;
    sel.luma = F.luma; sel.bpm = F.bpm;
    sel.tsel = (uint32_t)F.tab_dc[0] | ((uint32_t)F.tab_dc[1] << 4) | ((uint32_t)F.tab_dc[2] << 8) | ((uint32_t)F.tab_ac[0] << 12) |
               ((uint32_t)F.tab_ac[1] << 16) | ((uint32_t)F.tab_ac[2] << 20);
    // per-subsequence arrays of this frame: [0] segment, [1] entry bit, [2] entry phase, [3] exit bit, [4] exit phase,
    // [5] blocks completed, [6] candidate bit, [7] candidate phase, [8] exclusive block prefix
    const int MS = F.max_sub;
    uint32_t* S = sub_all + (size_t)F.sub_off;
    uint32_t *u_seg = S, *u_ebit = S + MS, *u_eph = S + 2 * (size_t)MS, *u_xbit = S + 3 * (size_t)MS, *u_xph = S + 4 * (size_t)MS,
             *u_nblk = S + 5 * (size_t)MS, *u_cbit = S + 6 * (size_t)MS, *u_cph = S + 7 * (size_t)MS, *u_pre = S + 8 * (size_t)MS;
    // first subsequence of every interval (prefix over the intervals' subsequence counts), kept behind the segment table
    uint32_t* sub_first = const_cast<uint32_t*>(sb) + F.max_seg;          // [max_seg + 1]
    if (t == 0) {
        int acc = 0;
        for (int g = 0; g < n_seg; ++g) {
            sub_first[g] = (uint32_t)acc;
            const uint32_t len = sb[g + 1] - sb[g];
            int n = (int)((len + JH_SB - 1) / JH_SB);
            if (n < 1) n = 1;
            acc += n;
        }
        sub_first[n_seg] = (uint32_t)acc;
        s_nsub = acc <= MS ? acc : -1;
        s_err = I->err;
        s_changed = 0;
        s_redo = 0;
    }
    __syncthreads();
    const int n_sub = s_nsub;
    if (n_sub < 0) { if (t == 0) { F.hdr.valid = 0; status[f] = LF_ERR_DECODE; } return; }
    for (int g = t; g < n_seg; g += JH_T)
        for (uint32_t u = sub_first[g]; u < sub_first[g + 1]; ++u) u_seg[u] = (uint32_t)g;
    __syncthreads();
    int dummy_err = 0;
    const long long jt0 = getenv_debug ? (long long)wall_clock64() : 0;     // (LF_JH_DEBUG: where a frame's time goes, x10 ns)
    // ---- pass 0: assumed entry states.  A decoder that starts out of step finds the bit position and the coefficient index
    // again within a block or two as long as it applies the right tables, i.e. knows its place in the MCU -- with the wrong place
    // it stays lost for ~25 subsequences (22 - 31 correction passes per frame until round 6, each a subsequence's latency with two
    // or three lanes at work).  So the scan is decoded once per PLACE: UNITS of `sup` consecutive subsequences, each from "block h
    // of the MCU, coefficient 0 at its first bit" for h = 0 .. bpm - 1, one lane per (unit, h), the state noted where it crosses
    // into each subsequence.  The place a unit is left in is a map h -> h': composing the maps along the interval (a scan) names the
    // hypothesis every unit is really entered with -- a unit is long enough for that hypothesis to have locked on before its end --
    // and the correction passes below start from its notes: the units' first subsequences (entered off by part of a symbol), then the
    // ~5 % whose predecessor had not locked on yet, then nothing.
    const int bpm = sel.bpm;
    uint32_t *h_bit = S + 9 * (size_t)MS, *h_pn = S + (9 + (size_t)bpm) * MS;      // [bpm][MS]: exit bit; exit phase | blocks completed << 16
    uint32_t* unit_q = S + (9 + 2 * (size_t)bpm) * MS;        // [MS]: a unit's hypothesis, kept at its first subsequence
    uint32_t* unit_lock = S + (10 + 2 * (size_t)bpm) * MS;    // [MS]: one thread per pass rewrites a unit's subsequences
    uint32_t* sub_pass = S + (11 + 2 * (size_t)bpm) * MS;     // [MS]: the pass (+ 1) a subsequence was last listed for decoding in
    const bool listed = n_sub <= JH_DL;                       // (larger scans: every subsequence its own unit)
    const int sup = listed ? ((n_sub * bpm + JH_T - 1) / JH_T > 1 ? (n_sub * bpm + JH_T - 1) / JH_T : 1) : 1;
    // the units' first subsequences, in order
    int n_units = n_sub;
    if (listed) {
        if (t == 0) s_carry = 0;
        __syncthreads();
        for (int base = 0; base < n_sub; base += JH_T) {
            const int u = base + t;
            const bool head = u < n_sub && ((uint32_t)u - sub_first[u_seg[u]]) % (uint32_t)sup == 0u;
            const unsigned long long bm = __ballot(head);
            if (lane == 0) s_wsum[wave] = __popcll(bm);
            __syncthreads();
            int off = s_carry;
            for (int w = 0; w < wave; ++w) off += s_wsum[w];
            if (head) sh.s_dlist[off + __popcll(bm & ((1ull << lane) - 1ull))] = (uint16_t)u;
            __syncthreads();
            if (t == JH_T - 1) s_carry = off + __popcll(bm);
            __syncthreads();
        }
        n_units = s_carry;
    }
    auto unit_head = [&](int k) { return listed ? (int)sh.s_dlist[k] : k; };
    for (int j = t; j < n_units * bpm; j += JH_T) {
        const int h = j / n_units, u0 = unit_head(j - h * n_units);
        const int g = (int)u_seg[u0];
        const uint32_t local0 = (uint32_t)u0 - sub_first[g];
        if (local0 == 0u && h != 0) continue;                 // an interval starts with block 0
        const uint32_t seg_end = sb[g + 1] * 8u;
        const int u_end = u0 + sup < (int)sub_first[g + 1] ? u0 + sup : (int)sub_first[g + 1];
        JhState st; st.bit = (sb[g] + local0 * JH_SB) * 8u; st.blk = h; st.k = 0;
        const int bad0 = dummy_err;
        for (int u = u0; u < u_end; ++u) {
            uint32_t limit = (sb[g] + ((uint32_t)u - sub_first[g] + 1u) * JH_SB) * 8u;
            if (limit > seg_end || (uint32_t)u + 1 == sub_first[g + 1]) limit = seg_end;
            const int nb = jh_span<false, LDS>(tabs, sel, clean, st, limit, seg_end, nullptr, 0, 0, dummy_err);
            h_bit[(size_t)h * MS + u] = st.bit; h_pn[(size_t)h * MS + u] = ((uint32_t)st.blk << 8) | (uint32_t)st.k | ((uint32_t)nb << 16) | ((uint32_t)(dummy_err - bad0 < 255 ? dummy_err - bad0 : 255) << 24);
        }
    }
    __syncthreads();
    {
        // maps as 4 bits per place (bpm <= 10, T.81 B.2.3); "a then b" = b[a[h]].  A thread composes its run of consecutive units,
        // the runs are scanned across the workgroup, and the thread walks its run again with the place it is entered in
        auto then = [&](unsigned long long a, unsigned long long b) {
            unsigned long long r = 0ull;
            for (int h = 0; h < bpm; ++h) r |= ((b >> (4 * (int)((a >> (4 * h)) & 15ull))) & 15ull) << (4 * h);
            return r;
        };
        auto unit_last = [&](int u0) { const int e = (int)sub_first[u_seg[u0] + 1]; return (u0 + sup < e ? u0 + sup : e) - 1; };
        auto map_of = [&](int u0) {
            unsigned long long m = 0ull;
            const bool first = (uint32_t)u0 == sub_first[u_seg[u0]];
            const int ul = unit_last(u0);
            for (int h = 0; h < bpm; ++h) m |= (unsigned long long)((h_pn[(size_t)(first ? 0 : h) * MS + ul] >> 8) & 15u) << (4 * h);
            return m;
        };
        unsigned long long ident = 0ull;
        for (int h = 0; h < bpm; ++h) ident |= (unsigned long long)h << (4 * h);
        const int R = (n_units + JH_T - 1) / JH_T, k0 = t * R < n_units ? t * R : n_units, k1 = k0 + R < n_units ? k0 + R : n_units;
        unsigned long long run = ident;
        for (int k = k0; k < k1; ++k) run = then(run, map_of(unit_head(k)));
        unsigned long long inc = run;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long o = (unsigned long long)__shfl_up((long long)inc, d);
            if (lane >= d) inc = then(o, inc);
        }
        if (lane == 63) sh.s_wmap[wave] = inc;
        __syncthreads();
        unsigned long long before = ident;
        for (int w = 0; w < wave; ++w) before = then(before, sh.s_wmap[w]);
        const unsigned long long up = (unsigned long long)__shfl_up((long long)inc, 1);
        if (lane > 0) before = then(before, up);
        int q = (int)(before & 15ull);                        // (the frame's first unit heads an interval: what comes before it is void)
        for (int k = k0; k < k1; ++k) {
            const int u0 = unit_head(k);
            if ((uint32_t)u0 == sub_first[u_seg[u0]]) q = 0;
            unit_q[u0] = (uint32_t)q;
            q = (int)((h_pn[(size_t)q * MS + unit_last(u0)] >> 8) & 15u);
        }
    }
    __syncthreads();
    // every subsequence: exit as its unit's hypothesis notes it; entry = what that hypothesis entered it with -- its predecessor's
    // exit inside a unit, and for a unit's first subsequence nothing a predecessor can hand over (so that the first pass decodes it)
    for (int u = t; u < n_sub; u += JH_T) {
        const int g = (int)u_seg[u];
        const uint32_t local = (uint32_t)u - sub_first[g], in_unit = local % (uint32_t)sup;
        const int q = (int)unit_q[u - (int)in_unit];
        const uint32_t pn = h_pn[(size_t)q * MS + u];
        u_xbit[u] = h_bit[(size_t)q * MS + u]; u_xph[u] = pn & 0xffffu; u_nblk[u] = (pn >> 16) & 255u;
        if (local == 0u) { u_ebit[u] = sb[g] * 8u; u_eph[u] = 0u; }
        else if (in_unit == 0u) { u_ebit[u] = 0xffffffffu; u_eph[u] = 0u; }
        else { u_ebit[u] = h_bit[(size_t)q * MS + u - 1]; u_eph[u] = h_pn[(size_t)q * MS + u - 1] & 0xffffu; }
    }
    __syncthreads();
    const long long jt1 = getenv_debug ? (long long)wall_clock64() : 0;
    // ---- until every subsequence starts from what its predecessor hands over
    if (n_sub <= JH_DL) {
        // The subsequences a pass has to decode again are COMPACTED first (round 5): after the first two or three passes they are a few
        // dozen, scattered over all sixteen waves -- and a wave with one live lane issues the symbol loop's ~100 instructions per symbol
        // like a full one, four such waves to a SIMD.  Listed densely they keep one or two waves busy, one per SIMD.
        // A subsequence whose entry is not what its predecessor hands over: inside a unit, the unit's hypotheses are looked at
        // first -- the one that crossed into it in exactly that state has decoded the rest of the unit already, its notes ARE the
        // decode from that state (no other hypothesis can have locked on: the next pass finds the unit's remaining subsequences in
        // step); a unit's first subsequence is decoded, and the rest of the unit takes the notes of the hypothesis of its new place.
        auto adopt = [&](int h, int u_from, int u_end) {      // subsequences u_from .. u_end - 1 as hypothesis h notes them (u_from inside the unit)
            for (int w = u_from; w < u_end; ++w) {
                const uint32_t pn = h_pn[(size_t)h * MS + w];
                u_ebit[w] = h_bit[(size_t)h * MS + w - 1]; u_eph[w] = h_pn[(size_t)h * MS + w - 1] & 0xffffu;
                u_xbit[w] = h_bit[(size_t)h * MS + w]; u_xph[w] = pn & 0xffffu; u_nblk[w] = (pn >> 16) & 255u;
            }
        };
        for (int pass = 0; pass < 2 * n_sub + 2; ++pass) {
            if (t == 0) { sh.s_nd = 0; s_changed = 0; }
            for (int u = t; u < n_sub; u += JH_T) unit_lock[u] = 0u;
            __syncthreads();
            for (int u0 = 0; u0 < n_sub; u0 += JH_T) {
                const int u = u0 + t;
                bool dirty = false;
                if (u < n_sub) {
                    const int g = (int)u_seg[u];
                    if ((uint32_t)u != sub_first[g]) {
                        const uint32_t cb = u_xbit[u - 1], cp = u_xph[u - 1];
                        dirty = !(cb == u_ebit[u] && cp == u_eph[u]);
                        if (dirty && sup > 1) {
                            const int in_unit = (int)(((uint32_t)u - sub_first[g]) % (uint32_t)sup);
                            const int u_end = u - in_unit + sup < (int)sub_first[g + 1] ? u - in_unit + sup : (int)sub_first[g + 1];
                            // (one thread per unit and pass writes beyond its own subsequence: a record is one decode's entry AND exit)
                            if (atomicCAS(&unit_lock[u - in_unit], 0u, 1u) != 0u) { }
                            else if (in_unit > 0) {
                                int m = -1;
                                for (int h = 0; h < bpm; ++h)
                                    if (h_bit[(size_t)h * MS + u - 1] == cb && (h_pn[(size_t)h * MS + u - 1] & 0xffffu) == cp) { m = h; break; }
                                if (m >= 0) { adopt(m, u, u_end); dirty = false; s_changed = 1; }
                            } else {
                                const int q = (int)(cp >> 8);
                                if (q < bpm && (uint32_t)q != unit_q[u]) { unit_q[u] = (uint32_t)q; adopt(q, u + 1, u_end); }
                            }
                        }
                        if (dirty) { u_cbit[u] = cb; u_cph[u] = cp; sub_pass[u] = (uint32_t)pass + 1u; }
                    }
                }
                const unsigned long long bm = __ballot(dirty);
                if (bm != 0ull) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&sh.s_nd, __popcll(bm));
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (dirty) sh.s_dlist[base + __popcll(bm & ((1ull << lane) - 1ull))] = (uint16_t)u;
                }
            }
            __syncthreads();
            const int n_dirty = sh.s_nd;
            if (n_dirty == 0 && s_changed == 0) { if (t == 0 && getenv_debug) printf("frame %d: n_sub %d redone %d passes %d\n", f, n_sub, s_redo, pass + 1); break; }
            if (getenv_debug && t == 0) s_redo += n_dirty;
            if (getenv_debug > 1 && t == 0 && f == 0) { printf("  pass %d sup %d dirty %d:", pass, sup, n_dirty); for (int j = 0; j < n_dirty && j < 24; ++j) printf(" %d", (int)sh.s_dlist[j]); printf("\n"); }
            for (int j = t; j < n_dirty; j += JH_T) {
                const int u = (int)sh.s_dlist[j];
                const int g = (int)u_seg[u];
                const uint32_t cb = u_cbit[u], cp = u_cph[u];
                const uint32_t local = (uint32_t)u - sub_first[g];
                const uint32_t start = (sb[g] + local * JH_SB) * 8u, seg_end = sb[g + 1] * 8u;
                uint32_t limit = start + JH_SB * 8u;
                if (limit > seg_end || (uint32_t)u + 1 == sub_first[g + 1]) limit = seg_end;
                JhState st; st.bit = cb; st.blk = (int)(cp >> 8); st.k = (int)(cp & 255u);
                u_ebit[u] = cb; u_eph[u] = cp;
                const int nb = jh_span<false, LDS>(tabs, sel, clean, st, limit, seg_end, nullptr, 0, 0, dummy_err);
                u_xbit[u] = st.bit; u_xph[u] = ((uint32_t)st.blk << 8) | (uint32_t)st.k; u_nblk[u] = (uint32_t)nb;
                // ... and on through the rest of the unit while nothing there is in step with it: the subsequence after it that is
                // entered in another state and not on this pass's list is looked up among the hypotheses (their notes from there to
                // the next listed subsequence) or decoded in turn -- a front that no hypothesis meets walks a unit in one pass
                if (sup > 1) {
                    const int in_unit = (int)(local % (uint32_t)sup);
                    const int ue = u - in_unit + sup < (int)sub_first[g + 1] ? u - in_unit + sup : (int)sub_first[g + 1];
                    for (int w = u + 1; w < ue && sub_pass[w] != (uint32_t)pass + 1u; ++w) {
                        const uint32_t xb = st.bit, xp = ((uint32_t)st.blk << 8) | (uint32_t)st.k;
                        if (u_ebit[w] == xb && u_eph[w] == xp) break;
                        int m = -1;
                        for (int h = 0; h < bpm; ++h)
                            if (h_bit[(size_t)h * MS + w - 1] == xb && (h_pn[(size_t)h * MS + w - 1] & 0xffffu) == xp) { m = h; break; }
                        if (m >= 0) {
                            int e = w + 1;
                            while (e < ue && sub_pass[e] != (uint32_t)pass + 1u) ++e;
                            adopt(m, w, e);
                            break;
                        }
                        uint32_t lim = (sb[g] + ((uint32_t)w - sub_first[g] + 1u) * JH_SB) * 8u;
                        if (lim > seg_end || (uint32_t)w + 1 == sub_first[g + 1]) lim = seg_end;
                        u_ebit[w] = xb; u_eph[w] = xp;
                        const int nbw = jh_span<false, LDS>(tabs, sel, clean, st, lim, seg_end, nullptr, 0, 0, dummy_err);
                        u_xbit[w] = st.bit; u_xph[w] = ((uint32_t)st.blk << 8) | (uint32_t)st.k; u_nblk[w] = (uint32_t)nbw;
                    }
                }
            }
            __syncthreads();
        }
    } else
    for (int pass = 0; pass < n_sub + 1; ++pass) {
        for (int u = t; u < n_sub; u += JH_T) {
            const int g = (int)u_seg[u];
            if ((uint32_t)u != sub_first[g]) { u_cbit[u] = u_xbit[u - 1]; u_cph[u] = u_xph[u - 1]; }
        }
        __syncthreads();
        bool changed = false;
        for (int u = t; u < n_sub; u += JH_T) {
            const int g = (int)u_seg[u];
            if ((uint32_t)u == sub_first[g]) continue;
            const uint32_t cb = u_cbit[u], cp = u_cph[u];
            if (cb == u_ebit[u] && cp == u_eph[u]) continue;
            changed = true;
            if (getenv_debug) atomicAdd(&s_redo, 1);
            const uint32_t local = (uint32_t)u - sub_first[g];
            const uint32_t start = (sb[g] + local * JH_SB) * 8u, seg_end = sb[g + 1] * 8u;
            uint32_t limit = start + JH_SB * 8u;
            if (limit > seg_end || (uint32_t)u + 1 == sub_first[g + 1]) limit = seg_end;
            JhState st; st.bit = cb; st.blk = (int)(cp >> 8); st.k = (int)(cp & 255u);
            u_ebit[u] = cb; u_eph[u] = cp;
            const int nb = jh_span<false, LDS>(tabs, sel, clean, st, limit, seg_end, nullptr, 0, 0, dummy_err);
            u_xbit[u] = st.bit; u_xph[u] = ((uint32_t)st.blk << 8) | (uint32_t)st.k; u_nblk[u] = (uint32_t)nb;
        }
        if (changed) s_changed = 1;
        __syncthreads();
        const int any = s_changed;
        __syncthreads();
        if (t == 0) s_changed = 0;
        __syncthreads();
        if (!any) { if (t == 0 && getenv_debug) printf("frame %d: n_sub %d redone %d passes %d\n", f, n_sub, s_redo, pass + 1); break; }
    }
    const long long jt2 = getenv_debug ? (long long)wall_clock64() : 0;
    if (getenv_debug > 2 && t == 0 && f == 0 && sup > 1) {
        // (diagnostic) per position in the unit: hypotheses that had slid over an undecodable symbol by then -- among those whose
        // notes ARE the truth there (exit equals the final exit), and among the others
        int tb[8] = {0}, tn[8] = {0}, wb[8] = {0}, wn[8] = {0};
        for (int u = 0; u < n_sub; ++u) {
            const int g = (int)u_seg[u];
            const uint32_t local = (uint32_t)u - sub_first[g];
            const int i = (int)(local % (uint32_t)sup) < 8 ? (int)(local % (uint32_t)sup) : 7;
            for (int h = 0; h < bpm; ++h) {
                if (local < (uint32_t)sup && h != 0) continue;
                const uint32_t pn = h_pn[(size_t)h * MS + u];
                const bool truth = h_bit[(size_t)h * MS + u] == u_xbit[u] && (pn & 0xffffu) == u_xph[u];
                if (truth) { ++tn[i]; tb[i] += (pn >> 24) != 0u; } else { ++wn[i]; wb[i] += (pn >> 24) != 0u; }
            }
        }
        for (int i = 0; i < 8; ++i) printf("  unit position %d: true notes %d (flagged %d), others %d (flagged %d)\n", i, tn[i], tb[i], wn[i], wb[i]);
    }
    // ---- blocks completed before every subsequence (exclusive prefix over the frame's subsequences)
    if (t == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n_sub; base += JH_T) {
        const int u = base + t;
        const int v = u < n_sub ? (int)u_nblk[u] : 0;
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int a = __shfl_up(inc, d); if (lane >= d) inc += a; }
        if (lane == 63) s_wsum[wave] = inc;
        __syncthreads();
        int off = s_carry;
        for (int w = 0; w < wave; ++w) off += s_wsum[w];
        if (u < n_sub) u_pre[u] = (uint32_t)(off + inc - v);
        __syncthreads();
        if (t == JH_T - 1) s_carry = off + inc;
        __syncthreads();
    }
    const long long jt3 = getenv_debug ? (long long)wall_clock64() : 0;
    // ---- the decode that counts: coefficients out, a sequential decoder's checks
    int16_t* coef = coef_all + (size_t)F.coef_base * 64;
    int err = 0;
    for (int u = t; u < n_sub; u += JH_T) {
        const int g = (int)u_seg[u];
        const uint32_t local = (uint32_t)u - sub_first[g];
        const uint32_t start = (sb[g] + local * JH_SB) * 8u, seg_end = sb[g + 1] * 8u;
        uint32_t limit = start + JH_SB * 8u;
        const bool last = (uint32_t)u + 1 == sub_first[g + 1];
        if (limit > seg_end || last) limit = seg_end;
        const int first_block = F.restart > 0 ? g * F.restart * F.bpm : 0;
        int end_block = F.restart > 0 ? (g + 1) * F.restart * F.bpm : F.hdr.nblocks;
        if (end_block > F.hdr.nblocks) end_block = F.hdr.nblocks;
        const int cur = first_block + (int)(u_pre[u] - u_pre[sub_first[g]]);
        JhState st; st.bit = u_ebit[u]; st.blk = (int)(u_eph[u] >> 8); st.k = (int)(u_eph[u] & 255u);
        const int nb = jh_span<true, LDS>(tabs, sel, clean, st, limit, seg_end, coef, cur, end_block, err);
        if (last && !err) {
            if (cur + nb < end_block) err |= 8;                             // the interval ran out before its last block (more data than
                                                                            // blocks is ignored, as a sequential decoder does)
            // T.81 E.2.4: an interval that a restart marker follows ends with fewer than 8 padding bits
            if (g + 1 < n_seg && seg_end - st.bit >= 8u) err |= 8;
            if (g + 1 == n_seg && end_block != F.hdr.nblocks) err |= 8;
        }
    }
    if (err) atomicOr(&s_err, err);
    __threadfence_block();
    __syncthreads();
    const long long jt4 = getenv_debug ? (long long)wall_clock64() : 0;
    if (getenv_debug && t == 0) printf("frame %d: pass 0 %lld, passes %lld, prefix %lld, final decode %lld (x10 ns)\n", f, jt1 - jt0, jt2 - jt1, jt3 - jt2, jt4 - jt3);
    if (t == 0) {
        I->n_sub = n_sub;
        I->err = s_err;
        if (s_err) { F.hdr.valid = 0; status[f] = LF_ERR_DECODE; }
    }
    if (s_err) return;
    // ---- DC prediction (T.81 F.2.2.1): DIFF -> DC per component, restarting with every interval.  One thread per MCU;
    // segmented scan inside a wave, the waves' carries chained through LDS
    {
        const int n_mcu = F.n_mcu, bpm = sel.bpm, luma = sel.luma, R = F.restart > 0 ? F.restart : n_mcu;
        if (t < 3) sh.s_dcc[0][t] = 0;
        __syncthreads();
        for (int base = 0; base < n_mcu; base += JH_T) {
            const int m = base + t;
            const bool act = m < n_mcu;
            int sum[3] = { 0, 0, 0 };
            if (act)
                for (int r = 0; r < bpm; ++r) sum[r < luma ? 0 : r - luma + 1] += (int)coef[((size_t)m * bpm + r) * 64];
            const bool head = act && (m % R == 0);
            int inc[3] = { sum[0], sum[1], sum[2] };
            bool flag = head;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int a0 = __shfl_up(inc[0], d), a1 = __shfl_up(inc[1], d), a2 = __shfl_up(inc[2], d);
                const bool fl = __shfl_up((int)flag, d) != 0;
                if (lane >= d) {
                    if (!flag) { inc[0] += a0; inc[1] += a1; inc[2] += a2; }
                    flag = flag || fl;
                }
            }
            if (lane == 63) { sh.s_dc[wave][0] = inc[0]; sh.s_dc[wave][1] = inc[1]; sh.s_dc[wave][2] = inc[2]; sh.s_dc[wave][3] = flag ? 1 : 0; }
            __syncthreads();
            if (t == 0) {
                // carry into wave w + 1 = what wave w ends with: its own total since its last interval start, on top of
                // the carry it received unless an interval started inside it
                for (int w = 0; w < JH_T / 64; ++w)
                    for (int c = 0; c < 3; ++c) sh.s_dcc[w + 1][c] = sh.s_dc[w][c] + (sh.s_dc[w][3] ? 0 : sh.s_dcc[w][c]);
            }
            __syncthreads();
            if (act) {
                int pred[3];
                for (int c = 0; c < 3; ++c) pred[c] = inc[c] - sum[c] + (flag ? 0 : sh.s_dcc[wave][c]);
                for (int r = 0; r < bpm; ++r) {
                    const int c = r < luma ? 0 : r - luma + 1;
                    int16_t* p = coef + ((size_t)m * bpm + r) * 64;
                    pred[c] = (int)((unsigned)pred[c] + (unsigned)(int)*p);
                    *p = (int16_t)pred[c];
                }
            }
            __syncthreads();
            if (t < 3) sh.s_dcc[0][t] = sh.s_dcc[JH_T / 64][t];
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(JH_TD) void k_jh_decode(jpeg::DevFrame* __restrict__ frames, const uint8_t* __restrict__ bytes, uint8_t* __restrict__ clean_all,
                                                    uint32_t* __restrict__ seg_begin_all, JhInfo* __restrict__ info,
                                                    uint32_t* __restrict__ sub_all, int16_t* __restrict__ coef_all, int* __restrict__ status, int getenv_debug,
                                                    int lds_clean_bytes, int unstuff_here)
{
    __shared__ JhShared sh;
    __shared__ JhUShared<JH_TD> shu;
    const int f = blockIdx.x, t = threadIdx.x;
    jpeg::DevFrame& F = frames[f];
    JhInfo* I = info + f;
    if (!F.hdr.valid) { if (unstuff_here && t == 0) { I->clean_len = 0; I->n_seg = 0; I->n_sub = 0; I->err = 0; } return; }
    uint8_t* clean = clean_all + F.clean_off;
    uint32_t* sb = seg_begin_all + F.seg_off;
    if (unstuff_here) {
        // the frame's raw scan through LDS (every scan of the batch fits: the host has checked), unstuffed into `clean` by all 1 024
        // threads: k_jh_unstuff's 256 took 0.17 ms per batch for it, and a launch
        const uint32_t L = F.scan_len;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(bytes + F.scan_off);     // scan_off is a multiple of 16
        const int n4 = (int)((L + 3) / 4);
        for (int i = t; i < n4; i += JH_TD) jh_dyn[i + (i >> 5)] = src[i];
        __syncthreads();
        jh_unstuff_body<true, JH_TD>(shu, F, I, reinterpret_cast<const uint8_t*>(jh_dyn), clean, sb);
        __threadfence_block();
        __syncthreads();
    }
    for (int i = t; i < (int)(sizeof(JhTabs) / 4); i += JH_TD) reinterpret_cast<uint32_t*>(&sh.tabs)[i] = reinterpret_cast<const uint32_t*>(F.tabs)[i];
    if (lds_clean_bytes > 0 && (I->clean_len + 48) + (I->clean_len + 48) / 32 + 16 <= lds_clean_bytes) {
        // the unstuffed scan into LDS once: every pass reads it again
        const uint32_t* src = reinterpret_cast<const uint32_t*>(clean);     // clean_off is a multiple of 16
        const int n4 = (I->clean_len + 32 + 3) / 4;
        for (int i = t; i < n4; i += JH_TD) jh_dyn[i + (i >> 5)] = src[i];
        jh_frame<true>(sh, F, I, clean, sb, sub_all, coef_all, status, getenv_debug, f);
    } else {
        jh_frame<false>(sh, F, I, clean, sb, sub_all, coef_all, status, getenv_debug, f);
    }
}

// ---------------------------------------------------------------------------------------------- k_jh_dc
// DC prediction (T.81 F.2.2.1): DIFF -> DC, per component, restarting at every interval.  One wave per frame walks the MCUs
// 64 at a time; the blocks of a component inside an MCU chain serially (at most four).
__global__ __launch_bounds__(64) void k_jh_dc(const jpeg::DevFrame* __restrict__ frames, int16_t* __restrict__ coef_all)
{
    const int f = blockIdx.x, lane = threadIdx.x;
    const jpeg::DevFrame& F = frames[f];
    if (!F.hdr.valid) return;
    int16_t* coef = coef_all + (size_t)F.coef_base * 64;
    const int n_mcu = F.n_mcu, bpm = F.bpm, luma = F.luma, R = F.restart > 0 ? F.restart : n_mcu;
    const int ncomp = F.hdr.ncomp;
    int carry[3] = { 0, 0, 0 };
    for (int base = 0; base < n_mcu; base += 64) {
        const int m = base + lane;
        const bool act = m < n_mcu;
        // sum of this MCU's differences per component
        int sum[3] = { 0, 0, 0 };
        if (act) {
            for (int r = 0; r < bpm; ++r) {
                const int c = r < luma ? 0 : r - luma + 1;
                sum[c] += (int)coef[((size_t)m * bpm + r) * 64];
            }
        }
        // segmented inclusive scan over the lanes: a lane whose MCU starts an interval does not take what is below it
        const bool head = act && (m % R == 0);
        int inc[3] = { sum[0], sum[1], sum[2] };
        bool flag = head;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int a0 = __shfl_up(inc[0], d), a1 = __shfl_up(inc[1], d), a2 = __shfl_up(inc[2], d);
            const bool fl = __shfl_up((int)flag, d) != 0;
            if (lane >= d) {
                if (!flag) { inc[0] += a0; inc[1] += a1; inc[2] += a2; }
                flag = flag || fl;
            }
        }
        // predictor before this MCU: the exclusive sum inside its interval, plus what the previous 64 MCUs carried over
        // unless an interval starts at or below this lane in this group
        int pred[3];
        for (int c = 0; c < 3; ++c) pred[c] = inc[c] - sum[c] + (flag ? 0 : carry[c]);
        if (act) {
            for (int r = 0; r < bpm; ++r) {
                const int c = r < luma ? 0 : r - luma + 1;
                int16_t* p = coef + ((size_t)m * bpm + r) * 64;
                pred[c] = (int)((unsigned)pred[c] + (unsigned)(int)*p);
                *p = (int16_t)pred[c];
            }
        }
        // carry = predictor after the last MCU of this group of 64
        const int lastl = (n_mcu - base < 64 ? n_mcu - base : 64) - 1;
        for (int c = 0; c < 3; ++c) carry[c] = __shfl(pred[c], lastl);
        (void)ncomp;
    }
}

// ---------------------------------------------------------------------------------------------- dense IDCT
namespace {
__device__ __forceinline__ int32_t jd_mulw(int32_t a, int32_t c) { return (int32_t)((uint32_t)a * (uint32_t)c); }
__device__ __forceinline__ int32_t jd_addw(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
__device__ __forceinline__ int32_t jd_subw(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
__device__ __forceinline__ void jd_idct8(const int32_t in[8], int32_t out[8], int shift)
{
    const int32_t z1 = jd_mulw(jd_addw(in[2], in[6]), 4433);
    const int32_t t2 = jd_addw(z1, jd_mulw(in[6], -15137));
    const int32_t t3 = jd_addw(z1, jd_mulw(in[2], 6270));
    const int32_t t0 = (int32_t)((uint32_t)jd_addw(in[0], in[4]) << 13);
    const int32_t t1 = (int32_t)((uint32_t)jd_subw(in[0], in[4]) << 13);
    const int32_t e0 = jd_addw(t0, t3), e3 = jd_subw(t0, t3), e1 = jd_addw(t1, t2), e2 = jd_subw(t1, t2);
    int32_t o0 = in[7], o1 = in[5], o2 = in[3], o3 = in[1];
    int32_t y1 = jd_addw(o0, o3), y2 = jd_addw(o1, o2), y3 = jd_addw(o0, o2), y4 = jd_addw(o1, o3);
    const int32_t y5 = jd_mulw(jd_addw(y3, y4), 9633);
    o0 = jd_mulw(o0, 2446); o1 = jd_mulw(o1, 16819); o2 = jd_mulw(o2, 25172); o3 = jd_mulw(o3, 12299);
    y1 = jd_mulw(y1, -7373); y2 = jd_mulw(y2, -20995);
    y3 = jd_addw(jd_mulw(y3, -16069), y5); y4 = jd_addw(jd_mulw(y4, -3196), y5);
    o0 = jd_addw(o0, jd_addw(y1, y3)); o1 = jd_addw(o1, jd_addw(y2, y4));
    o2 = jd_addw(o2, jd_addw(y2, y3)); o3 = jd_addw(o3, jd_addw(y1, y4));
    const int32_t r = (int32_t)1 << (shift - 1);
    out[0] = jd_addw(jd_addw(e0, o3), r) >> shift; out[7] = jd_addw(jd_subw(e0, o3), r) >> shift;
    out[1] = jd_addw(jd_addw(e1, o2), r) >> shift; out[6] = jd_addw(jd_subw(e1, o2), r) >> shift;
    out[2] = jd_addw(jd_addw(e2, o1), r) >> shift; out[5] = jd_addw(jd_subw(e2, o1), r) >> shift;
    out[3] = jd_addw(jd_addw(e3, o0), r) >> shift; out[4] = jd_addw(jd_subw(e3, o0), r) >> shift;
}
__device__ __forceinline__ uint32_t jd_limit(int32_t x)
{
    const int idx = (int)((uint32_t)x & 1023u);
    return idx < 128 ? (uint32_t)(128 + idx) : idx < 512 ? 255u : idx < 896 ? 0u : (uint32_t)(idx - 896);
}
constexpr int kDBlocksPerWg = 32, kDStride = 72;
}

// the dequantisation + inverse DCT of k_jpeg_idct (k_jpeg.hip), reading dense blocks: 8 threads per block, thread k loads
// row k (16 bytes) and owns column k in pass 1, row k in pass 2
__global__ __launch_bounds__(256) void k_jpeg_idct_dense(JpegGeom g, const jpeg::DevFrame* __restrict__ frames, const int16_t* __restrict__ coef_all,
                                                         uint8_t* __restrict__ planes)
{
    __shared__ int32_t ws[kDBlocksPerWg * kDStride];
    const int f = blockIdx.y;
    const jpeg::DevFrame& F = frames[f];
    const int t = threadIdx.x, lb = t >> 3, k = t & 7;
    // rows above g.first_row are not wanted: the MCU rows that lie wholly more than 16 rows above it (fancy upsampling looks one
    // chroma row up) are the frame's first blocks -- skipped as whole workgroups
    const int skip_rows = F.hdr.valid && g.first_row > 16 ? (g.first_row - 16) / (8 * F.hdr.vmax) : 0;
    const int b0 = skip_rows * F.hdr.mcux * F.bpm;
    const int b = b0 + blockIdx.x * kDBlocksPerWg + lb;
    const int nblocks = F.hdr.valid ? F.hdr.nblocks : 0;
    if (b0 + (int)blockIdx.x * kDBlocksPerWg >= nblocks) return;
    const bool act = b < nblocks;
    int32_t* w = ws + lb * kDStride;
    int comp = 0, bx = 0, by = 0;
    int32_t in[8], res[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) in[c] = 0;
    if (act) {
        const int hm = F.hdr.hmax, vm = F.hdr.vmax;
        const int luma = F.luma, bpm = F.bpm;
        const int mcu = b / bpm, r = b - mcu * bpm;
        const int my = mcu / F.hdr.mcux, mx = mcu - my * F.hdr.mcux;
        if (r < luma) { comp = 0; bx = mx * hm + (r % hm); by = my * vm + (r / hm); }
        else { comp = r - luma + 1; bx = mx; by = my; }
        const uint16_t* q = F.hdr.qt[comp] + 8 * k;
        const uint4 v = *reinterpret_cast<const uint4*>(coef_all + ((size_t)F.coef_base + b) * 64 + 8 * k);
        const uint32_t wv[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
        for (int c = 0; c < 8; ++c) in[c] = jd_mulw((int32_t)(int16_t)((wv[c >> 1] >> (16 * (c & 1))) & 0xffffu), (int32_t)q[c]);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) w[k * 8 + c] = in[c];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) in[r] = w[r * 8 + k];
    jd_idct8(in, res, 13 - 2);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) w[r * 8 + k] = res[r];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 8; ++c) in[c] = w[k * 8 + c];
    jd_idct8(in, res, 13 + 2 + 3);
    if (act) {
        const uint32_t lo = jd_limit(res[0]) | (jd_limit(res[1]) << 8) | (jd_limit(res[2]) << 16) | (jd_limit(res[3]) << 24);
        const uint32_t hi = jd_limit(res[4]) | (jd_limit(res[5]) << 8) | (jd_limit(res[6]) << 16) | (jd_limit(res[7]) << 24);
        uint8_t* pl = planes + ((size_t)f * 3 + comp) * g.Hp * g.Wp;
        *reinterpret_cast<uint2*>(pl + (size_t)(by * 8 + k) * g.Wp + bx * 8) = make_uint2(lo, hi);
    }
}

void launch_jh_decode(const JpegGeom& g, int n_frames, int max_blocks, size_t max_scan_len, jpeg::DevFrame* frames, const uint8_t* bytes, uint8_t* clean,
                      uint32_t* seg_begin, void* info, uint32_t* sub, int16_t* coef, int* status, uint8_t* planes, hipStream_t s)
{
    JhInfo* I = static_cast<JhInfo*>(info);
    int lds_raw = (int)(max_scan_len + max_scan_len / 32 + 64 + 15) & ~15;
    if (lds_raw > JH_LDS_CLEAN) lds_raw = 0;                             // larger scans are read from global memory
    static const int dbg = getenv("LF_JH_DEBUG") ? atoi(getenv("LF_JH_DEBUG")) : 0;          // diagnostic: passes until the subsequences agree, per frame
    static const int split = getenv("LF_JH_SPLIT") ? 1 : 0;                                    // A/B: the unstuffing kernel of rounds 3 - 5 in front
    // LDS for the largest scan of the batch (clean <= raw), up to JH_LDS_CLEAN; larger scans are read from global memory
    int lds = (int)(max_scan_len + 64 + (max_scan_len + 64) / 32 + 32 + 15) & ~15;
    if (lds > JH_LDS_CLEAN) lds = JH_LDS_CLEAN;
    const int unstuff_here = lds_raw > 0 && lds_raw <= lds && !split;    // every raw scan of the batch fits the decoder's LDS
    if (!unstuff_here) {
        if (lds_raw > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_jh_unstuff), hipFuncAttributeMaxDynamicSharedMemorySize, lds_raw);
        hipLaunchKernelGGL(k_jh_unstuff, dim3(n_frames), dim3(JH_T), (size_t)lds_raw, s, frames, bytes, clean, seg_begin, I, lds_raw);
    }
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_jh_decode), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k_jh_decode, dim3(n_frames), dim3(JH_TD), (size_t)lds, s, frames, bytes, clean, seg_begin, I, sub, coef, status, dbg, lds, unstuff_here);
    // (DC prediction happens at the end of k_jh_decode; k_jh_dc is the stand-alone form, kept for scans decoded elsewhere)
    if (max_blocks > 0) {
        const dim3 grid((unsigned)((max_blocks + kDBlocksPerWg - 1) / kDBlocksPerWg), (unsigned)n_frames);
        hipLaunchKernelGGL(k_jpeg_idct_dense, grid, dim3(256), 0, s, g, frames, coef, planes);
    }
}

size_t jh_info_bytes() { return sizeof(JhInfo); }

}  // namespace lf
