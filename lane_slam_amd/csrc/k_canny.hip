// K_canny: cv2.Canny(bgr, lo, hi, apertureSize=3) on the 3-channel working image.
//
// Reference call site: /root/reference/src/line_detector/include/line_detector/line_detector_lsd.py:60-62,139.
// Arithmetic (OpenCV 3.x canny.cpp, restated): Sobel 3x3 -> s16 per channel with
// BORDER_REPLICATE, L1 magnitude, per pixel the channel with the largest magnitude (first
// wins ties), non-maximum suppression with the 15-bit TG22 sector test against a
// zero-padded magnitude image, then 8-connected hysteresis.
//
// k_canny_nms: one 64x16 tile per workgroup; BGR tile (+2 halo) -> LDS, magnitude tile
//   (+1 halo) -> LDS, NMS classification; each wave owns image rows so the 64-lane ballot
//   IS the 64-pixel bit-plane word pair.  Output: `weak` (m > low and local max) and
//   `strong` (also m > high) bit planes, 1 bit per pixel -- 8x fewer bytes than u8 maps.
//   Algorithmic bytes per pixel: 3 read (moved as one BGRX dword), 2/8 written.
// k_hysteresis: one workgroup per frame, both bit planes resident in LDS (k_hysteresis_strips: strip by strip
//   for frames that are too large); Jacobi sweeps of
//   "strong |= weak & dilate3x3(strong)" with an exact in-word run fill (carry trick) until
//   a sweep changes nothing.  The fixpoint is unique, so the result does not depend on
//   sweep order (== the reference's stack-based flood fill).
#include "common.h"

namespace lf {

constexpr int CT_W = 64, CT_H = 16;

__global__ __launch_bounds__(256) void k_canny_nms(CannyParams p, const uint32_t* __restrict__ bgr,
                                                   uint32_t* __restrict__ strong, uint32_t* __restrict__ weak)
{
    // the working image's pixels, unpacked ONCE when they are loaded: B | R << 16 (two 16-bit lanes for packed
    // arithmetic) and G.  Every magnitude pixel reads its eight neighbours from both planes: no byte extraction in
    // the 8-fold reused inner formula.
    __shared__ uint32_t pbr[(CT_H + 4) * (CT_W + 4)];
    __shared__ uint32_t pg[(CT_H + 4) * (CT_W + 4)];
    __shared__ int mag[(CT_H + 2) * (CT_W + 2)];
    __shared__ int gxy[(CT_H + 2) * (CT_W + 2)];
    int tbx, tby, f;
    lf_xcd_tile(tbx, tby, f);
    const int x0 = tbx * CT_W, y0 = tby * CT_H;
    const uint32_t* img = bgr + (size_t)f * p.Hc * p.W;
    constexpr int PW = CT_W + 4, PH = CT_H + 4, MW = CT_W + 2, MH = CT_H + 2;
    typedef short s2 __attribute__((ext_vector_type(2)));

    const int tid = threadIdx.y * 64 + threadIdx.x;
    for (int idx = tid; idx < PW * PH; idx += 256) {
        const int ty = idx / PW, tx = idx - ty * PW;
        const int gx = min(max(x0 + tx - 2, 0), p.W - 1);
        const int gy = min(max(y0 + ty - 2, 0), p.Hc - 1);
        const uint32_t a = img[(size_t)gy * p.W + gx];
        pbr[idx] = a & 0x00ff00ffu;
        pg[idx] = (a >> 8) & 0xffu;
    }
    __syncthreads();
    {
        for (int idx = tid; idx < MW * MH; idx += 256) {
            const int ty = idx / MW, tx = idx - ty * MW;
            const int gx = x0 + tx - 1, gy = y0 + ty - 1;
            int best = 0, bx = 0, by = 0;
            if (gx >= 0 && gx < p.W && gy >= 0 && gy < p.Hc) {
                const int o = (ty + 1) * PW + (tx + 1);
                // B and R together, two signed 16-bit lanes (|dx|, |dy| <= 1020)
                const uint32_t* c = pbr + o;
                const s2 v00 = __builtin_bit_cast(s2, c[-PW - 1]), v01 = __builtin_bit_cast(s2, c[-PW]), v02 = __builtin_bit_cast(s2, c[-PW + 1]);
                const s2 v10 = __builtin_bit_cast(s2, c[-1]), v12 = __builtin_bit_cast(s2, c[1]);
                const s2 v20 = __builtin_bit_cast(s2, c[PW - 1]), v21 = __builtin_bit_cast(s2, c[PW]), v22 = __builtin_bit_cast(s2, c[PW + 1]);
                const s2 two = { 2, 2 }, zero = { 0, 0 };
                const s2 dx2 = (v02 - v00) + (v22 - v20) + two * (v12 - v10);
                const s2 dy2 = (v20 - v00) + (v22 - v02) + two * (v21 - v01);
                const s2 ax2 = __builtin_elementwise_max(dx2, zero - dx2), ay2 = __builtin_elementwise_max(dy2, zero - dy2);
                const s2 m2 = ax2 + ay2;
                // G in plain integers
                const uint32_t* g = pg + o;
                const int g00 = (int)g[-PW - 1], g01 = (int)g[-PW], g02 = (int)g[-PW + 1], g10 = (int)g[-1], g12 = (int)g[1];
                const int g20 = (int)g[PW - 1], g21 = (int)g[PW], g22 = (int)g[PW + 1];
                const int dxg = (g02 - g00) + 2 * (g12 - g10) + (g22 - g20);
                const int dyg = (g20 - g00) + 2 * (g21 - g01) + (g22 - g02);
                const int mg = abs(dxg) + abs(dyg);
                // channel order B, G, R; the first of equal magnitudes wins
                best = (int)m2.x; bx = (int)dx2.x; by = (int)dy2.x;
                if (mg > best) { best = mg; bx = dxg; by = dyg; }
                if ((int)m2.y > best) { best = (int)m2.y; bx = (int)dx2.y; by = (int)dy2.y; }
            }
            mag[idx] = best;
            gxy[idx] = (bx & 0xFFFF) | (by << 16);
        }
    }
    __syncthreads();
    const int TG22 = (int)(0.4142135623730950488016887242097 * (1 << 15) + 0.5);
    for (int ry = threadIdx.y; ry < CT_H; ry += 4) {
        const int gx = x0 + threadIdx.x, gy = y0 + ry;
        bool keep = false, hi = false;
        if (gx < p.W && gy < p.Hc) {
            const int* pm = mag + (ry + 1) * MW + threadIdx.x + 1;
            int m = pm[0];
            if (m > p.low) {
                int g = gxy[(ry + 1) * MW + threadIdx.x + 1];
                int xs = (int)(short)(g & 0xFFFF), ys = g >> 16;
                int ax = abs(xs);
                int ay = abs(ys) << 15;
                int tg22x = ax * TG22;
                if (ay < tg22x) keep = (m > pm[-1] && m >= pm[1]);
                else {
                    int tg67x = tg22x + (ax << 16);
                    if (ay > tg67x) keep = (m > pm[-MW] && m >= pm[MW]);
                    else {
                        int s = (xs ^ ys) < 0 ? -1 : 1;
                        keep = (m > pm[-MW - s] && m > pm[MW + s]);
                    }
                }
                hi = keep && m > p.high;
            }
        }
        unsigned long long bw = __ballot(keep), bs = __ballot(hi);
        if (gy < p.Hc) {
            int word = (x0 >> 5) + (threadIdx.x >> 5);
            if ((threadIdx.x & 31) == 0 && word < p.Ww) {
                size_t o = ((size_t)f * p.Hc + gy) * p.Ww + word;
                int sh = threadIdx.x & 32;
                weak[o] = (uint32_t)(bw >> sh);
                strong[o] = (uint32_t)(bs >> sh);
            }
        }
    }
}

void launch_canny(const CannyParams& p, const uint32_t* bgr, int n_frames, uint32_t* strong, uint32_t* weak,
                  hipStream_t s)
{
    dim3 grid((p.W + CT_W - 1) / CT_W, (p.Hc + CT_H - 1) / CT_H, n_frames);
    hipLaunchKernelGGL(k_canny_nms, grid, dim3(64, 4), 0, s, p, bgr, strong, weak);
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
    if (lds <= 64 * 1024 && nw <= 8 * 1024) {
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
