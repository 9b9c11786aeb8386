// K_canny: cv2.Canny(bgr, lo, hi, apertureSize=3) on the 3-channel working image.
//
// Reference call site: /root/reference/src/line_detector/include/line_detector/line_detector_lsd.py:60-62,139.
// Arithmetic (OpenCV 3.x canny.cpp, restated): Sobel 3x3 -> s16 per channel with
// BORDER_REPLICATE, L1 magnitude, per pixel the channel with the largest magnitude (first
// wins ties), non-maximum suppression with the 15-bit TG22 sector test against a
// zero-padded magnitude image, then 8-connected hysteresis.
//
// k_canny_nms: one wave per 64-column x 62-row band, lane = column, rows streamed through registers (see the kernel);
//   the 64-lane ballot of a row IS the 64-pixel bit-plane word pair.  Output: `weak` (m > low and local max) and
//   `strong` (also m > high) bit planes, 1 bit per pixel -- 8x fewer bytes than u8 maps.
//   Algorithmic bytes per pixel: 3 read (moved as one BGRX dword), 2/8 written.
// k_hysteresis: one workgroup per frame, both bit planes resident in LDS (k_hysteresis_strips: strip by strip
//   for frames that are too large); Jacobi sweeps of
//   "strong |= weak & dilate3x3(strong)" with an exact in-word run fill (carry trick) until
//   a sweep changes nothing.  The fixpoint is unique, so the result does not depend on
//   sweep order (== the reference's stack-based flood fill).
#include "common.h"

namespace lf {

// ---- k_canny_nms: one WAVE per 64-column x 62-row band, lane = column, rows stream through registers ----------
// The kernel is bound by vector instructions, not by bytes (the tile version spent 159 lane-operations per pixel:
// 16 LDS reads per magnitude, index arithmetic, two LDS round trips), so the layout is chosen for instruction count:
//   * a lane owns one image column and walks down the rows; per row it loads its pixel and its two horizontal
//     neighbours (three coalesced dword loads through the texture cache, columns clamped = BORDER_REPLICATE), and
//     forms the row's horizontal difference and smoothing terms ONCE (Sobel is separable): B and R together in two
//     16-bit lanes, G beside them;
//   * a pixel's dx, dy combine the terms of three consecutive rows, which rotate through three register sets (the
//     loop body is three steps, so nothing is ever copied); rows are fetched three steps ahead of their use;
//   * per row the only unconditional work is the three magnitudes and their maximum; everything else -- which channel
//     won, the sector test, the comparison with the neighbours -- runs only when some pixel of the 64 exceeds the low
//     threshold, behind one wave-uniform branch;
//   * horizontal neighbours of the magnitudes come from the adjacent lanes (DPP wave shifts); the two columns next to
//     the strip (x0 - 1 and x0 + 64) are worked out with lane = (side, row), 18 rows at a time in step with the main
//     loop, and enter the shifts as the value of the vacated lane;
//   * the 64-lane ballot of a row IS the pair of bit-plane words, as before.
#ifndef LF_CANNY_ROWS
#define LF_CANNY_ROWS 62
#endif
constexpr int CB_R = LF_CANNY_ROWS;
static_assert(CB_R + 2 <= 64, "the columns beside a strip are worked out with one lane per magnitude row");

typedef short cn_s2 __attribute__((ext_vector_type(2)));

struct CannyH {            // horizontal terms of one pixel row at this lane's column
    cn_s2 d_br, s_br;      // right - left, left + 2 centre + right: (B, R)
    int d_g, s_g;          // the same for G
};
struct CannyG {            // gradient of one pixel, all channels, and the channel magnitudes
    cn_s2 dx_br, dy_br;
    int dx_g, dy_g;
    uint32_t m_br;         // |dx| + |dy|: B in the low half, R in the high half
    int m_g;
};

// a + 2 b in both 16-bit lanes, one instruction (the compiler splits it into a shift and an add)
__device__ __forceinline__ cn_s2 canny_a_plus_2b(cn_s2 a, cn_s2 b)
{
    cn_s2 d;
    const uint32_t two = 0x00020002u;
    asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(d) : "v"(b), "v"(two), "v"(a));
    return d;
}

__device__ __forceinline__ CannyH canny_hterms(uint32_t aL, uint32_t aM, uint32_t aR)
{
    const cn_s2 l = __builtin_bit_cast(cn_s2, aL & 0x00ff00ffu), m = __builtin_bit_cast(cn_s2, aM & 0x00ff00ffu), r = __builtin_bit_cast(cn_s2, aR & 0x00ff00ffu);
    const int lg = (int)((aL >> 8) & 0xffu), mg = (int)((aM >> 8) & 0xffu), rg = (int)((aR >> 8) & 0xffu);
    CannyH h;
    h.d_br = r - l;
    h.s_br = canny_a_plus_2b(l, m) + r;
    h.d_g = rg - lg;
    h.s_g = lg + 2 * mg + rg;
    return h;
}

// largest channel magnitude at (x, y), 0 outside the image (the zero-padded magnitude plane); plain integers, used
// for the two columns beside a strip only
__device__ __forceinline__ int canny_mag_at(const uint32_t* __restrict__ img, int W, int Hc, int x, int y)
{
    if (x < 0 || x >= W || y < 0 || y >= Hc) return 0;
    const int xl = max(x - 1, 0), xr = min(x + 1, W - 1), yu = max(y - 1, 0), yd = min(y + 1, Hc - 1);
    const uint32_t a00 = img[(size_t)yu * W + xl], a01 = img[(size_t)yu * W + x], a02 = img[(size_t)yu * W + xr];
    const uint32_t a10 = img[(size_t)y * W + xl], a12 = img[(size_t)y * W + xr];
    const uint32_t a20 = img[(size_t)yd * W + xl], a21 = img[(size_t)yd * W + x], a22 = img[(size_t)yd * W + xr];
    int best = 0;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const int sh = 8 * ch;
        const int v00 = (a00 >> sh) & 255, v01 = (a01 >> sh) & 255, v02 = (a02 >> sh) & 255, v10 = (a10 >> sh) & 255, v12 = (a12 >> sh) & 255;
        const int v20 = (a20 >> sh) & 255, v21 = (a21 >> sh) & 255, v22 = (a22 >> sh) & 255;
        const int dx = (v02 - v00) + 2 * (v12 - v10) + (v22 - v20);
        const int dy = (v20 - v00) + 2 * (v21 - v01) + (v22 - v02);
        best = max(best, abs(dx) + abs(dy));
    }
    return best;
}

// lane i takes v of lane i - 1 (i + 1); the vacated lane 0 (63) takes `edge`
__device__ __forceinline__ int canny_from_left(int v, int edge) { return __builtin_amdgcn_update_dpp(edge, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int canny_from_right(int v, int edge) { return __builtin_amdgcn_update_dpp(edge, v, 0x130, 0xf, 0xf, false); }

struct CannyBand {
    const uint32_t* img;       // this frame's working image
    __amdgpu_buffer_rsrc_t rsrc;   // ... as a raw buffer
    __amdgpu_buffer_rsrc_t rs_weak, rs_strong;   // this frame's bit planes
    int W, Hc, Ww, low, high;
    int x0, y0, y_end;         // band: rows y0 .. y_end - 1
    uint32_t xl, xm, xr;       // this lane's clamped columns, as byte offsets into a row
    int lane;
    bool in_image;             // this lane's column exists
    bool st_ok;                // this lane stores a word of the bit planes (lanes 0, 1: the strip's first / second word)
    uint32_t st_off;           // ... at this byte offset from the strip's first word
    int* halo;                 // LDS, this wave's: [j] = magnitude at (x0 - 1, y0 - 1 + j), [64 + j] = at (x0 + 64, y0 - 1 + j)
};

__device__ __forceinline__ void canny_fetch(const CannyBand& c, int prow, uint32_t (&px)[3])
{
    const int yy = min(max(prow, 0), c.Hc - 1);
    // buffer loads: scalar row offset + the lane's fixed byte offset, no per-lane address arithmetic
    const int row = yy * c.W * 4;
    px[0] = __builtin_amdgcn_raw_buffer_load_b32(c.rsrc, (int)c.xl, row, 0);
    px[1] = __builtin_amdgcn_raw_buffer_load_b32(c.rsrc, (int)c.xm, row, 0);
    px[2] = __builtin_amdgcn_raw_buffer_load_b32(c.rsrc, (int)c.xr, row, 0);
}

// The two columns beside the strip, 18 rows at a time with lane = (side, row): step t decides row y0 + t - 4 and needs the
// magnitudes of rows j = t - 4 .. t - 2 (j counted from y0 - 1), so the chunk computed at step t = 16 k covers
// j = 16 k - 4 .. 16 k + 13.  Doing this for the whole band up front read one 128-byte line per row and side long before
// the neighbouring strip's wave streams through those lines: with a dozen frames in flight per L2 they were evicted
// in between and fetched twice (PMC: 2.07 x the algorithmic bytes; 1.12 x without the pass).  In step with the rows
// the lines are the ones the neighbour is reading at that moment.
__device__ __forceinline__ void canny_halo_chunk(const CannyBand& c, int t)
{
    const int side = c.lane >> 5, j = t - 4 + (c.lane & 31);
    if ((c.lane & 31) < 18 && j >= 0 && j < 64)
        c.halo[side * 64 + j] = canny_mag_at(c.img, c.W, c.Hc, side ? c.x0 + 64 : c.x0 - 1, c.y0 - 1 + j);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// one row step.  Slots: (A, B, C) = the register sets of pixel rows (p - 2, p - 1, p); set C is overwritten here.
template <int A, int B, int C>
__device__ __forceinline__ void canny_step(const CannyBand& c, int t, uint32_t (&px)[3][3], CannyH (&H)[3], int (&M)[3], CannyG (&G)[3])
{
    const int p = c.y0 - 2 + t;
    if ((t & 15) == 0) canny_halo_chunk(c, t);
    H[C] = canny_hterms(px[C][0], px[C][1], px[C][2]);
    canny_fetch(c, p + 3, px[C]);                       // three rows ahead
    if (t < 2) return;
    // magnitudes of row y = p - 1
    {
        const int y = p - 1;
        const cn_s2 zero = { 0, 0 };
        CannyG g;
        g.dx_br = canny_a_plus_2b(H[A].d_br, H[B].d_br) + H[C].d_br;
        g.dy_br = H[C].s_br - H[A].s_br;
        g.dx_g = H[A].d_g + 2 * H[B].d_g + H[C].d_g;
        g.dy_g = H[C].s_g - H[A].s_g;
        const cn_s2 m_br = __builtin_elementwise_max(g.dx_br, zero - g.dx_br) + __builtin_elementwise_max(g.dy_br, zero - g.dy_br);
        g.m_g = abs(g.dx_g) + abs(g.dy_g);
        g.m_br = __builtin_bit_cast(uint32_t, m_br);
        const int best = max(max((int)(g.m_br & 0xffffu), (int)(g.m_br >> 16)), g.m_g);
        M[C] = (c.in_image && y >= 0 && y < c.Hc) ? best : 0;
        G[C] = g;
    }
    const int yn = p - 2;                               // magnitudes M[A] (yn - 1), M[B] (yn), M[C] (yn + 1)
    if (yn < c.y0 || yn >= c.y_end) return;
    const int m = M[B];
    // lane decisions are kept as 64-bit lane masks: the sector logic is scalar bit arithmetic and the result IS the output
    const unsigned long long over = __ballot(m > c.low);
    unsigned long long keep = 0ull, hi = 0ull;
    if (over != 0ull) {
        // the magnitudes beside the strip: one LDS word per row and side, the same for every lane (it only lands in lane 0 / 63)
        const int* hl = c.halo + (yn - c.y0 + 1);
        const int up = M[A], dn = M[C];
        const int mL = canny_from_left(m, hl[0]), mR = canny_from_right(m, hl[64]);
        const int upL = canny_from_left(up, hl[-1]), upR = canny_from_right(up, hl[63]);
        const int dnL = canny_from_left(dn, hl[1]), dnR = canny_from_right(dn, hl[65]);
        // channel order B, G, R; the first of equal magnitudes wins
        const CannyG g = G[B];
        const int m_b = (int)(g.m_br & 0xffffu), m_r = (int)(g.m_br >> 16);
        const bool take_g = g.m_g > m_b;
        const bool take_r = m_r > max(m_b, g.m_g);
        const int xs = take_r ? (int)g.dx_br.y : (take_g ? g.dx_g : (int)g.dx_br.x);
        const int ys = take_r ? (int)g.dy_br.y : (take_g ? g.dy_g : (int)g.dy_br.x);
        const int TG22 = (int)(0.4142135623730950488016887242097 * (1 << 15) + 0.5);
        const int ax = abs(xs), ay = abs(ys) << 15;          // ax <= 1020: the 24-bit multiply is exact
        int tg22x;
        asm("v_mul_u32_u24 %0, %1, %2" : "=v"(tg22x) : "v"(ax), "v"(TG22));        // full rate; the compiler picks the quarter-rate 32-bit multiply
        const int tg67x = tg22x + (ax << 16);
        const bool neg = (xs ^ ys) < 0;                      // s = -1: (x + 1, y - 1) and (x - 1, y + 1)
        const unsigned long long keep_h = __ballot(m > mL) & __ballot(m >= mR);
        const unsigned long long keep_v = __ballot(m > up) & __ballot(m >= dn);
        const unsigned long long keep_d = __ballot(m > (neg ? upR : upL)) & __ballot(m > (neg ? dnL : dnR));
        const unsigned long long is_h = __ballot(ay < tg22x), is_v = __ballot(ay > tg67x) & ~is_h;
        keep = over & ((is_h & keep_h) | (is_v & keep_v) | (~(is_h | is_v) & keep_d));
        hi = keep & __ballot(m > c.high);
    }
    // lanes 0 and 1 store the row's two words of each plane (buffer stores: scalar row offset + the lane's fixed offset)
    if (c.st_ok) {
        const int o = (yn * c.Ww + (c.x0 >> 5)) * 4;
        const bool second = c.lane != 0;
        __builtin_amdgcn_raw_buffer_store_b32(second ? (uint32_t)(keep >> 32) : (uint32_t)keep, c.rs_weak, (int)c.st_off, o, 0);
        __builtin_amdgcn_raw_buffer_store_b32(second ? (uint32_t)(hi >> 32) : (uint32_t)hi, c.rs_strong, (int)c.st_off, o, 0);
    }
}

#ifndef LF_CANNY_WAVES
#define LF_CANNY_WAVES 7
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LF_CANNY_WAVES, LF_CANNY_WAVES))) void k_canny_nms(CannyParams p, int n_units, int n_strips, int n_bands,
                                                   const uint32_t* __restrict__ bgr, uint32_t* __restrict__ strong,
                                                   uint32_t* __restrict__ weak)
{
    __shared__ int halo[4][128];
    // Workgroups go round the eight XCDs (each with its own L2) in launch order.  The strips and bands of one frame share
    // pixel columns and rows at their borders (and the lane = row pass above reads a line per row of the neighbouring
    // strip), so a frame's workgroups are all given to ONE XCD: block b runs on XCD b % 8 and takes a workgroup of frame
    // 8 * (group of eight frames) + b % 8.  Frames past the last full group of eight keep the launch order.
    int b = (int)blockIdx.x;
    {
        const int per_frame = (n_strips * n_bands + 3) / 4;          // workgroups per frame (units padded to whole workgroups)
        const int full = (n_units / (n_strips * n_bands)) / 8 * 8 * per_frame;
        if (b < full) {
            const int grp = b / (8 * per_frame), r = b - grp * 8 * per_frame;
            b = (grp * 8 + (r & 7)) * per_frame + (r >> 3);
        }
    }
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // wave-uniform: rows, bounds and row addresses stay in scalar registers
    const int per_frame_u = n_strips * n_bands, wg_per_frame = (per_frame_u + 3) / 4;
    const int f = b / wg_per_frame, uf = (b - f * wg_per_frame) * 4 + wave;         // unit inside the frame
    if (uf >= per_frame_u || f * per_frame_u >= n_units) return;                    // no barriers in this kernel: waves are independent
    const int u = uf;
    CannyBand c;
    c.lane = (int)(threadIdx.x & 63u);
    const int strip = u % n_strips, band = u / n_strips;
    c.W = p.W; c.Hc = p.Hc; c.Ww = p.Ww; c.low = p.low; c.high = p.high;
    c.img = bgr + (size_t)f * p.Hc * p.W;
    c.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(c.img), 0, p.Hc * p.W * 4, 0x00020000);
    c.rs_weak = __builtin_amdgcn_make_buffer_rsrc(weak + (size_t)f * p.Hc * p.Ww, 0, p.Hc * p.Ww * 4, 0x00020000);
    c.rs_strong = __builtin_amdgcn_make_buffer_rsrc(strong + (size_t)f * p.Hc * p.Ww, 0, p.Hc * p.Ww * 4, 0x00020000);
    c.x0 = strip * 64; c.y0 = band * CB_R; c.y_end = min(c.y0 + CB_R, p.Hc);
    const int x = c.x0 + c.lane;
    c.in_image = x < p.W;
    c.st_ok = c.lane < 2 && (c.x0 >> 5) + c.lane < p.Ww;
    c.st_off = 4u * (uint32_t)c.lane;
    const int xm = min(x, p.W - 1);
    c.xm = 4u * (uint32_t)xm; c.xl = 4u * (uint32_t)max(xm - 1, 0); c.xr = 4u * (uint32_t)min(xm + 1, p.W - 1);
    // the two columns beside the strip, lane = row (rows y0 - 1 .. y0 + 62)
    c.halo = halo[threadIdx.x >> 6];
    uint32_t px[3][3];
    CannyH H[3];
    int M[3] = { 0, 0, 0 };
    CannyG G[3];
    canny_fetch(c, c.y0 - 2, px[0]);
    canny_fetch(c, c.y0 - 1, px[1]);
    canny_fetch(c, c.y0, px[2]);
    const int steps = (c.y_end - c.y0) + 4;             // pixel rows y0 - 2 .. y_end + 1; output row yn is decided at step yn - y0 + 4
    for (int t = 0; t < steps; t += 3) {
        canny_step<1, 2, 0>(c, t, px, H, M, G);
        canny_step<2, 0, 1>(c, t + 1, px, H, M, G);
        canny_step<0, 1, 2>(c, t + 2, px, H, M, G);
    }
}

void launch_canny(const CannyParams& p, const uint32_t* bgr, int n_frames, uint32_t* strong, uint32_t* weak,
                  hipStream_t s)
{
    const int n_strips = (p.W + 63) / 64, n_bands = (p.Hc + CB_R - 1) / CB_R;
    const int n_units = n_strips * n_bands * n_frames;
    const int wg_per_frame = (n_strips * n_bands + 3) / 4;
    hipLaunchKernelGGL(k_canny_nms, dim3(wg_per_frame * n_frames), dim3(256), 0, s, p, n_units, n_strips, n_bands, bgr, strong, weak);
}

// fill every run of ones in w that contains a one of s (s is a subset of w), upward direction
__device__ __forceinline__ uint32_t fill_up(uint32_t w, uint32_t s) { return (((w + s) ^ w) & w) | s; }

__global__ __launch_bounds__(1024) void k_hysteresis(CannyParams p, uint32_t* __restrict__ strong,
                                                      const uint32_t* __restrict__ weak)
{
    extern __shared__ uint32_t lds[];
    const int nw = p.Hc * p.Ww;
    uint32_t* S = lds;
    uint32_t* Wk = lds + nw;
    const int f = blockIdx.x;
    uint32_t* gs = strong + (size_t)f * nw;
    const uint32_t* gw = weak + (size_t)f * nw;
    for (int i = threadIdx.x; i < nw; i += blockDim.x) { S[i] = gs[i]; Wk[i] = gw[i]; }
    __syncthreads();
    const int Ww = p.Ww;
    for (int iter = 0; iter < 65536; ++iter) {
        int changed = 0;
        // Jacobi sweep: read S (old), write into registers, then store after a barrier
        uint32_t upd[8];
#pragma unroll
        for (int cnt = 0; cnt < 8; ++cnt) {
            const int i = threadIdx.x + cnt * 1024;
            upd[cnt] = 0;
            if (i >= nw) continue;
            int y = i / Ww, x = i - y * Ww;
            uint32_t w = Wk[i];
            uint32_t cur = S[i];
            uint32_t acc = 0;
            if (w) {
#pragma unroll
                for (int dy = -1; dy <= 1; ++dy) {
                    int yy = y + dy;
                    if (yy < 0 || yy >= p.Hc) continue;
                    const uint32_t* row = S + yy * Ww;
                    uint32_t c = row[x];
                    uint32_t l = x > 0 ? row[x - 1] : 0u;
                    uint32_t r = x + 1 < Ww ? row[x + 1] : 0u;
                    acc |= c | (c << 1) | (c >> 1) | (l >> 31) | (r << 31);
                }
                uint32_t s = acc & w;
                // exact horizontal run fill inside the word, both directions
                s = fill_up(w, s);
                s = __brev(fill_up(__brev(w), __brev(s)));
                acc = s;
            }
            upd[cnt] = acc | cur;
            changed |= (upd[cnt] != cur);
        }
        __syncthreads();
#pragma unroll
        for (int cnt = 0; cnt < 8; ++cnt) {
            const int i = threadIdx.x + cnt * 1024;
            if (i < nw) S[i] = upd[cnt];
        }
        if (!__syncthreads_or(changed)) break;
    }
    for (int i = threadIdx.x; i < nw; i += blockDim.x) gs[i] = S[i];
}

// K_hysteresis_cols (round 4): the same fixpoint, reached in far fewer sweeps on real images.  k_hysteresis moves a strong mark one
// row per sweep (and any distance inside a 32-pixel word), so a weak chain running down the image costs as many sweeps as it has
// rows: 19 x the time of the lane frames on camera frames (1.19 ms per batch, a third of everything that is not region growing
// there).  Here a thread owns R consecutive rows of one word column, holds them in registers and sweeps them down and then up
// inside one iteration, each row seeing the rows already updated: a mark crosses the thread's R rows in one iteration.  The words
// of the neighbouring columns and of the rows above and below the strip are read from LDS as they are at that moment -- other
// threads may or may not have written theirs yet: every value ever stored is a sound one (a weak pixel 8-connected to a strong
// one), marks are only ever added, and the loop ends after an iteration in which nobody stored anything, i.e. in which every
// thread saw the final state -- the unique fixpoint, whatever the interleaving.  One barrier per iteration.
// R = 32 rows per thread and 256 threads where that covers the image (640 x 320: 200 threads): a workgroup of four waves and 25 KB
// of LDS is what one workgroup of k_lsd_grow leaves behind on a CU -- with 1024 threads (R = 8) the kernel waited 7 - 17 ms for
// a CU with sixteen free wave slots while other batches' region growing held 24 of 32 everywhere (tools/pipe_overlap.py).
template <int R, int NT>
__global__ __launch_bounds__(NT) void k_hysteresis_cols(CannyParams p, uint32_t* __restrict__ strong, const uint32_t* __restrict__ weak)
{
    extern __shared__ uint32_t lds[];
    const int nw = p.Hc * p.Ww, Ww = p.Ww, Hc = p.Hc;
    uint32_t* S = lds;
    const int f = blockIdx.x;
    uint32_t* gs = strong + (size_t)f * nw;
    const uint32_t* gw = weak + (size_t)f * nw;
    for (int i = threadIdx.x; i < nw; i += blockDim.x) S[i] = gs[i];
    const int t = threadIdx.x;
    const int x = t % Ww, y0 = (t / Ww) * R;
    const bool mine = y0 < Hc;
    uint32_t w[R], cur[R];
    bool any_weak = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        w[r] = (mine && y0 + r < Hc) ? gw[(y0 + r) * Ww + x] : 0u;
        any_weak = any_weak || w[r] != 0u;
    }
    __syncthreads();
    auto spread = [](uint32_t c) { return c | (c << 1) | (c >> 1); };
    for (int iter = 0; iter < 65536; ++iter) {
        int changed = 0;
        if (any_weak) {
            // rows y0 - 1 .. y0 + R of this column; of the two columns beside it only the bit that touches this one matters (bit 31
            // of the left word, bit 0 of the right one): one bit per row, kept as two masks (bit r + 1 = row y0 + r)
            uint32_t above = 0u, below = 0u;
            unsigned long long sideL = 0ull, sideR = 0ull;
            int at = (y0 - 1) * Ww + x;                        // walks down the column; opaque to the optimiser, which would otherwise
            int yv = y0, xv = x;                               // keep 102 loop-invariant addresses and as many lane masks in registers
            asm volatile("" : "+v"(at), "+v"(yv), "+v"(xv));   // across the iterations (256 VGPRs, 190 spilled SGPRs)
#pragma unroll
            for (int r = -1; r <= R; ++r, at += Ww) {
                const int yy = yv + r;
                const bool in = yy >= 0 && yy < Hc;
                const uint32_t* row = S + (in ? at - xv : 0);
                const uint32_t c = in ? row[xv] : 0u;
                const uint32_t l = in && xv > 0 ? row[xv - 1] : 0u;
                const uint32_t rr = in && xv + 1 < Ww ? row[xv + 1] : 0u;
                sideL |= (unsigned long long)(l >> 31) << (r + 1);
                sideR |= (unsigned long long)(rr & 1u) << (r + 1);
                if (r == -1) above = c; else if (r == R) below = c; else { cur[r] = c; asm volatile("" : "+v"(w[r])); }   // (nor anything derived from w[r])
                if ((r & 7) == 6) asm volatile("" ::: "memory");      // eight rows' reads in flight at a time, not all 34 (registers: see above)
            }
            uint32_t dirty = 0u;                               // rows of the strip that changed (R <= 32)
            auto step = [&](int r) {
                if (w[r] == 0u) return;
                const uint32_t up = r == 0 ? above : cur[r - 1], dn = r == R - 1 ? below : cur[r + 1];
                const uint32_t side = (((sideL >> r) & 7ull) ? 1u : 0u) | (((sideR >> r) & 7ull) ? 0x80000000u : 0u);   // rows r - 1, r, r + 1
                uint32_t sgot = (spread(up) | spread(cur[r]) | spread(dn) | side) & w[r];
                sgot = fill_up(w[r], sgot);
                sgot = __brev(fill_up(__brev(w[r]), __brev(sgot)));
                if (sgot & ~cur[r]) { cur[r] |= sgot; dirty |= 1u << r; }
            };
#pragma unroll
            for (int r = 0; r < R; ++r) { step(r); asm volatile("" : "+v"(w[r])); }      // (the up sweep recomputes what it needs from w[r])
#pragma unroll
            for (int r = R - 2; r >= 0; --r) step(r);
#pragma unroll
            for (int r = 0; r < R; ++r)
                if ((dirty >> r) & 1u) S[at - (R + 1 - r) * Ww] = cur[r];
            changed = dirty != 0u;
        }
        if (!__syncthreads_or(changed)) break;
    }
    for (int i = threadIdx.x; i < nw; i += blockDim.x) gs[i] = S[i];
}

// Working images whose two bit planes do not fit one workgroup's LDS (e.g. 1920x720: 2 x 173 KB): one workgroup
// per frame sweeps the frame in horizontal STRIPS.  A strip (its weak rows, its strong rows and one strong halo row
// above and below, which stay fixed) is brought into LDS, iterated to its local fixpoint exactly as k_hysteresis
// does, and written back; strips are visited top to bottom, then bottom to top, until a whole down + up cycle
// changes nothing.  Strong bits only ever get set and the fixpoint is unique, so the visiting order is free;
// a chain of weak pixels crosses any number of strips in one pass in its direction of travel.
__global__ __launch_bounds__(1024) void k_hysteresis_strips(CannyParams p, int strip_rows, uint32_t* __restrict__ strong,
                                                             const uint32_t* __restrict__ weak)
{
    extern __shared__ uint32_t lds[];
    const int Ww = p.Ww, Hc = p.Hc;
    const int f = blockIdx.x;
    uint32_t* gs = strong + (size_t)f * Hc * Ww;
    const uint32_t* gw = weak + (size_t)f * Hc * Ww;
    uint32_t* S = lds;                                   // [(strip_rows + 2)][Ww], row 0 / last = halo
    uint32_t* Wk = lds + (size_t)(strip_rows + 2) * Ww;  // [strip_rows][Ww]
    const int n_strips = (Hc + strip_rows - 1) / strip_rows;
    for (int cycle = 0; cycle < 65536; ++cycle) {
        int cycle_changed = 0;
        for (int pass = 0; pass < 2; ++pass) {
            for (int k = 0; k < n_strips; ++k) {
                const int st = pass == 0 ? k : n_strips - 1 - k;
                const int r0 = st * strip_rows, nr = min(strip_rows, Hc - r0);
                const int nw = nr * Ww;
                __syncthreads();
                for (int i = threadIdx.x; i < (nr + 2) * Ww; i += 1024) {
                    const int ry = i / Ww - 1 + r0;                       // image row of this LDS row
                    S[i] = (ry >= 0 && ry < Hc) ? gs[(size_t)ry * Ww + (i % Ww)] : 0u;
                }
                for (int i = threadIdx.x; i < nw; i += 1024) Wk[i] = gw[(size_t)r0 * Ww + i];
                __syncthreads();
                int strip_changed = 0;
                for (int iter = 0; iter < 65536; ++iter) {
                    int changed = 0;
                    uint32_t upd[8];
#pragma unroll
                    for (int cnt = 0; cnt < 8; ++cnt) {
                        const int i = threadIdx.x + cnt * 1024;
                        upd[cnt] = 0;
                        if (i >= nw) continue;
                        const int y = i / Ww, x = i - y * Ww;
                        const uint32_t w = Wk[i];
                        const uint32_t cur = S[(y + 1) * Ww + x];
                        uint32_t acc = 0;
                        if (w) {
#pragma unroll
                            for (int dy = 0; dy <= 2; ++dy) {                  // LDS rows y .. y+2 = image rows y-1 .. y+1
                                const uint32_t* row = S + (y + dy) * Ww;
                                const uint32_t c = row[x];
                                const uint32_t l = x > 0 ? row[x - 1] : 0u;
                                const uint32_t r = x + 1 < Ww ? row[x + 1] : 0u;
                                acc |= c | (c << 1) | (c >> 1) | (l >> 31) | (r << 31);
                            }
                            uint32_t sfill = acc & w;
                            sfill = fill_up(w, sfill);
                            sfill = __brev(fill_up(__brev(w), __brev(sfill)));
                            acc = sfill;
                        }
                        upd[cnt] = acc | cur;
                        changed |= (upd[cnt] != cur);
                    }
                    __syncthreads();
#pragma unroll
                    for (int cnt = 0; cnt < 8; ++cnt) {
                        const int i = threadIdx.x + cnt * 1024;
                        if (i < nw) { const int y = i / Ww, x = i - y * Ww; S[(y + 1) * Ww + x] = upd[cnt]; }
                    }
                    if (!__syncthreads_or(changed)) break;
                    strip_changed = 1;
                }
                if (strip_changed) {
                    for (int i = threadIdx.x; i < nw; i += 1024) gs[(size_t)r0 * Ww + i] = S[Ww + i];
                    cycle_changed = 1;
                }
                __threadfence_block();
            }
        }
        if (!__syncthreads_or(cycle_changed)) break;
    }
}

int launch_hysteresis(const CannyParams& p, int n_frames, uint32_t* strong, const uint32_t* weak, hipStream_t s)
{
    const int nw = p.Hc * p.Ww;
    const size_t lds = (size_t)nw * 2 * sizeof(uint32_t);
    static const bool jacobi = getenv("LF_HYST_JACOBI") != nullptr;         // A/B: the sweep-per-row kernel of rounds 1 - 3
    const bool fits = !jacobi && (size_t)nw * 4 <= 64 * 1024;
    if (fits && p.Ww * ((p.Hc + 31) / 32) <= 256) {
        hipLaunchKernelGGL((k_hysteresis_cols<32, 256>), dim3(n_frames), dim3(256), (size_t)nw * 4, s, p, strong, weak);
    } else if (fits && p.Ww * ((p.Hc + 15) / 16) <= 512) {
        hipLaunchKernelGGL((k_hysteresis_cols<16, 512>), dim3(n_frames), dim3(512), (size_t)nw * 4, s, p, strong, weak);
    } else if (fits && p.Ww * ((p.Hc + 15) / 16) <= 1024) {
        hipLaunchKernelGGL((k_hysteresis_cols<16, 1024>), dim3(n_frames), dim3(1024), (size_t)nw * 4, s, p, strong, weak);
    } else if (lds <= 64 * 1024 && nw <= 8 * 1024) {
        hipLaunchKernelGGL(k_hysteresis, dim3(n_frames), dim3(1024), lds, s, p, strong, weak);
    } else {
        // strips of at most 8192 words (8 per thread) and 60 KB of LDS for the three row sets
        int rows = 8192 / p.Ww;
        const int by_lds = (int)((60 * 1024 / sizeof(uint32_t)) / (2 * (size_t)p.Ww)) - 1;
        if (rows > by_lds) rows = by_lds;
        if (rows < 1) return -1;
        const size_t slds = ((size_t)(rows + 2) + rows) * p.Ww * sizeof(uint32_t);
        hipLaunchKernelGGL(k_hysteresis_strips, dim3(n_frames), dim3(1024), slds, s, p, rows, strong, weak);
    }
    return 0;
}

__global__ void k_edges_u8(CannyParams p, int n_frames, const uint32_t* __restrict__ bits, uint8_t* __restrict__ edges)
{
    size_t total = (size_t)n_frames * p.Hc * p.W;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t row = i / p.W;
        int x = (int)(i - row * p.W);
        uint32_t w = bits[row * p.Ww + (x >> 5)];
        edges[i] = ((w >> (x & 31)) & 1u) ? 255 : 0;
    }
}

__global__ void k_bgrx_to_bgr(int n_pix, const uint32_t* __restrict__ bgrx, uint8_t* __restrict__ bgr)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)n_pix; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t v = bgrx[i];
        bgr[3 * i] = (uint8_t)v; bgr[3 * i + 1] = (uint8_t)(v >> 8); bgr[3 * i + 2] = (uint8_t)(v >> 16);
    }
}

void launch_bgrx_to_bgr(int n_pix, const uint32_t* bgrx, uint8_t* bgr, hipStream_t s)
{
    hipLaunchKernelGGL(k_bgrx_to_bgr, dim3(1024), dim3(256), 0, s, n_pix, bgrx, bgr);
}

void launch_edges_u8(const CannyParams& p, int n_frames, const uint32_t* bits, uint8_t* edges, hipStream_t s)
{
    hipLaunchKernelGGL(k_edges_u8, dim3(1024), dim3(256), 0, s, p, n_frames, bits, edges);
}

}  // namespace lf
