// K_lsd_grad: first half of cv2 LSD (a-4) for every (frame, colour) at once:
//   edge_color = dilated colour mask & Canny edges        (line_detector_lsd.py:56)
//   -> f64 Gaussian blur -> bilinear resize by `scale` -> 2x2 gradient, level-line angle.
// Restates OpenCV 3.x lsd.cpp (flsd/ll_angle) and the filters it calls, in the exact
// operation order of the CPU oracle (oracle/lf_oracle_lsd.c):
//   cv::RowFilter<double>:        s = k[0]*S[0]; s += k[j]*S[j]
//   cv::SymmColumnFilter<double>: s = k[h]*S[0] + 0; s += k[h+j]*(S[j] + S[-j])
//   cv::resize INTER_LINEAR/f64:  float taps, double accumulation, horizontal then vertical
//   gradient: gx=(D-A)+(B-C), gy=(D-A)-(B-C), norm=sqrt((gx^2+gy^2)/4), fastAtan2(gx,-gy)
//
// The scaled image is never materialised in HBM, and neither are dense angle / magnitude
// planes: ~96 % of a colour's LSD image has no gradient at all.  Pass 1 ANDs the two bit planes
// under every 32x32 tile's footprint and lists the tiles that contain edge pixels (reads
// 2 bits per working pixel, writes a few KB).  Pass 2 runs the arithmetic for the listed tiles
// in LDS and appends one RECORD per pixel whose gradient is defined (address, angle, magnitude,
// cos/sin of the float-rounded angle) to the problem's record list; k_lsd_order sorts the
// records into raster order.  HBM traffic is proportional to the number of edge pixels.
#include "common.h"

namespace lf {

constexpr int GT = 32;   // scaled tile edge

__device__ __forceinline__ int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) { p = p < 0 ? -p : 2 * (n - 1) - p; }
    return p;
}

// idx / n for idx < 4096, n < 128 without a division: (idx * (2^19 / n + 1)) >> 19 (checked exhaustively); the tile
// phases below are flat loops over (row, column) pairs, so that all 256 lanes work whatever the tile's width is
// (41 or 42 raw columns and 33 output columns left 36 - 48 % of the lanes of a 64-wide row loop idle)
__device__ __forceinline__ int div_small(int idx, uint32_t magic) { return (int)(((uint32_t)idx * magic) >> 19); }

// ---- pass 1: list the tiles whose raw footprint contains edge_color pixels ------------------
// One workgroup per strip of 32 scaled rows: every edge-word AND mask-word under the strip's raw
// footprint is tested once; non-zero words mark their word column in LDS; a tile is listed iff a
// marked column lies under its footprint.
__global__ __launch_bounds__(256) void k_lsd_classify(LsdParams p, ResizeTables rt, const uint32_t* __restrict__ edge_bits,
                                                     const uint32_t* __restrict__ mask_bits, uint32_t* __restrict__ list,
                                                     int* __restrict__ list_count)
{
    __shared__ uint32_t colnz[256];
    __shared__ uint8_t tflag[256];
    const int h = p.half;
    const int pc = blockIdx.y, f = pc / 3, tid = threadIdx.x;
    const int ty = blockIdx.x, Y0 = ty * GT;
    const int Y1 = min(Y0 + GT, p.Hs - 1);
    const int sy_lo = rt.y0[Y0], sy_hi = rt.y1[Y1];
    // BORDER_REFLECT_101 only folds indices back inside these clamped ranges
    const int r0 = max(0, sy_lo - h), r1 = min(p.Hc - 1, sy_hi + h);
    const int Ww = p.Ww;
    for (int i = tid; i < Ww; i += 256) colnz[i] = 0;
    __syncthreads();
    const uint32_t* eb = edge_bits + (size_t)f * p.Hc * Ww;
    const uint32_t* mb = mask_bits + (size_t)pc * p.Hc * Ww;
    const int total = (r1 - r0 + 1) * Ww;
    for (int i = tid; i < total; i += 256) {
        const size_t o = (size_t)r0 * Ww + i;
        if (eb[o] & mb[o]) colnz[i % Ww] = 1;              // edge_color = dilated mask & edges
    }
    __syncthreads();
    const int ntx = (p.Ws + GT - 1) / GT;
    for (int tx = tid; tx < ntx; tx += 256) {
        const int X0 = tx * GT, X1 = min(X0 + GT, p.Ws - 1);
        const int sx_lo = rt.xofs[X0], sx_hi = min(rt.xofs[X1] + 1, p.W - 1);
        const int w0 = max(0, sx_lo - h) >> 5, w1 = min(p.W - 1, sx_hi + h) >> 5;
        uint32_t any = 0;
        for (int w = w0; w <= w1; ++w) any |= colnz[w];
        tflag[tx] = (uint8_t)any;
        if (any) {
            int slot = atomicAdd(list_count, 1);
            list[slot] = ((uint32_t)pc << 16) | ((uint32_t)ty << 8) | (uint32_t)tx;
        }
    }
}

// ---- pass 2: blur + resample + gradient for the listed tiles (persistent workgroups) ---------
// The raw tile is binary, so cv::RowFilter's ordered sum s = k[0]*S[0]; s += k[j]*S[j] only ever
// adds the constants k[j]*255 for set pixels (adding +0.0 is exact): each row-filter output is a
// lookup in a 2^ntaps-entry table indexed by the bits under the taps, and the raw tile lives in
// LDS as one 64-bit window per row.  Angles/sines are evaluated after compacting the tile's defined
// pixels so the expensive double-precision path runs on full waves.
#ifndef LF_LSD_GRAD_THREADS
#define LF_LSD_GRAD_THREADS 512
#endif
constexpr int LG_T = LF_LSD_GRAD_THREADS;      // threads per tile
#ifndef LF_GRAD_DIAG
#define LF_GRAD_DIAG 0          // timing experiments only: bits 1 / 2 / 4 / 8 leave out the column filter / the two resizes / the row filter
#endif
#ifndef LF_GRAD_WAVES
#define LF_GRAD_WAVES 0
#endif
#if LF_GRAD_WAVES
__attribute__((amdgpu_waves_per_eu(LF_GRAD_WAVES, LF_GRAD_WAVES)))
#endif
__global__ __launch_bounds__(LG_T) void k_lsd_grad(LsdParams p, ResizeTables rt, const uint32_t* __restrict__ edge_bits,
                                                  const uint32_t* __restrict__ mask_bits, uint32_t* __restrict__ r_addr,
                                                  float* __restrict__ r_deg, double* __restrict__ r_mod,
                                                  double* __restrict__ r_cs, double* __restrict__ r_sn,
                                                  int* __restrict__ n_rec, unsigned long long* __restrict__ maxgrad,
                                                  int max_nsx, int max_nsy, const uint32_t* __restrict__ list,
                                                  const int* __restrict__ list_count,
                                                  uint32_t* __restrict__ l_addr, double* __restrict__ l_mod, int* __restrict__ n_low,
                                                  const uint8_t* __restrict__ gray, int* __restrict__ rec_need)
{
    extern __shared__ double lds_d[];
    __shared__ double T[128];                     // ordered partial sums of k[j]*255 per 7-bit pattern
    __shared__ unsigned long long rowbits[GT * 2 + 2 * kMaxGaussTaps];
    __shared__ int n_def, rec_base, n_lo, low_base;
    __shared__ unsigned long long tile_max;
    __shared__ int t_xofs[GT + 1], t_y0[GT + 1], t_y1[GT + 1];
    __shared__ float t_xa[2 * (GT + 1)], t_yb[2 * (GT + 1)];
    const int h = p.half;
    const int n_tiles = *list_count;
    const bool use_table = p.ntaps <= 7;
    if (use_table) {
        for (int m = threadIdx.x; m < (1 << p.ntaps); m += LG_T) {
            double s = 0.0;
            bool first = true;
            for (int j = 0; j < p.ntaps; ++j) {
                const double term = p.k[j] * ((m >> j) & 1 ? 255.0 : 0.0);
                if (first) { s = term; first = false; } else s += term;
            }
            T[m] = s;
        }
    }
    const size_t szF = (size_t)(max_nsy + 2 * h) * max_nsx;
    const size_t szHb = (size_t)max_nsy * (GT + 1);
    size_t regA = szF > szHb ? szF : szHb;
    // also hosts the defined-pixel list (8 + 8 B entries) and, for the OpenCV >= 3.2 seed order (k_lsd_seed32.hip), the list of the
    // pixels whose gradient is NOT defined but not zero either (8 + 4 B entries): std::sort there orders every pixel
    const size_t min_a = (size_t)(l_addr ? 4 : 2) * GT * GT;
    if (regA < min_a) regA = min_a;
    double* F = lds_d;                                   // [rh][nsx]     row-filtered
    double* Bl = lds_d + regA;                           // [nsy][nsx]    blurred
    double* Hb = lds_d;                                  // [nsy][GT+1]   h-resized   (reuses F)
    double* Sc = lds_d + regA;                           // [GT+1][GT+1]  v-resized   (reuses Bl)
    uint2* dl = reinterpret_cast<uint2*>(lds_d);         // defined pixels (address, angle): reuses F/Hb once Sc is built
    double* dln = lds_d + (size_t)GT * GT;               // their gradient magnitudes
    double* lln = lds_d + (size_t)2 * GT * GT;           // "low" pixels: magnitude, address
    uint32_t* lla = reinterpret_cast<uint32_t*>(lds_d + (size_t)3 * GT * GT);
    const double DEG_TO_RADS = 3.14159265358979323846 / 180;

    for (int ti = blockIdx.x; ti < n_tiles; ti += gridDim.x) {
        __syncthreads();                           // LDS reuse across tiles
        const uint32_t tile = list[ti];
        const int pc = (int)(tile >> 16);          // problem = frame*3 + colour
        const int f = pc / 3;
        const int X0 = (int)(tile & 255u) * GT, Y0 = (int)((tile >> 8) & 255u) * GT;
        const int X1 = min(X0 + GT, p.Ws - 1), Y1 = min(Y0 + GT, p.Hs - 1);
        const int sx_lo = rt.xofs[X0], sx_hi = min(rt.xofs[X1] + 1, p.W - 1);
        const int sy_lo = rt.y0[Y0], sy_hi = rt.y1[Y1];
        const int nsx = sx_hi - sx_lo + 1, nsy = sy_hi - sy_lo + 1;
        const int nox = X1 - X0 + 1, noy = Y1 - Y0 + 1;   // scaled samples needed (incl. +1 neighbour)
        const int rw = nsx + 2 * h, rh = nsy + 2 * h;
        // gray != nullptr: LSD of a GRAY image (LSDDetectorC::detect runs cv's LSD on the levels of a gray pyramid,
        // LSDDetector_custom.cpp:150-160): the raw tile is the image itself, [frame][Hc][W] u8, no bit planes
        const uint8_t* gimg = gray ? gray + (size_t)f * p.Hc * p.W : nullptr;
        const uint32_t* mk = gray ? nullptr : mask_bits + (size_t)pc * p.Hc * p.Ww;
        const uint32_t* eb = gray ? nullptr : edge_bits + (size_t)f * p.Hc * p.Ww;
        if (threadIdx.x == 0) { n_def = 0; n_lo = 0; tile_max = 0ull; }
        // this tile's slice of the resize tables -> LDS (no dependent global loads in the passes below)
        if (threadIdx.x < nox) {
            const int dx = X0 + threadIdx.x;
            t_xofs[threadIdx.x] = rt.xofs[dx] - sx_lo;
            t_xa[2 * threadIdx.x] = rt.xa[2 * dx];
            t_xa[2 * threadIdx.x + 1] = rt.xa[2 * dx + 1];
        } else if (threadIdx.x >= 64 && threadIdx.x < 64 + noy) {
            const int oy = threadIdx.x - 64, dy = Y0 + oy;
            t_y0[oy] = rt.y0[dy] - sy_lo;
            t_y1[oy] = rt.y1[dy] - sy_lo;
            t_yb[2 * oy] = rt.yb[2 * dy];
            t_yb[2 * oy + 1] = rt.yb[2 * dy + 1];
        }
        // raw rows as bit windows: bit t of rowbits[ty] = edge_color(reflect(sx_lo-h+t), reflect(sy_lo-h+ty))
        const int xs = sx_lo - h;
        const bool interior_x = xs >= 0 && xs + rw <= p.W && rw <= 64;
        for (int ty = threadIdx.x; ty < (gray ? 0 : rh); ty += LG_T) {
            const int gy = reflect101(sy_lo - h + ty, p.Hc);
            const uint32_t* er = eb + (size_t)gy * p.Ww;
            const uint32_t* mr = mk + (size_t)gy * p.Ww;
            unsigned long long bits = 0;
            if (interior_x) {
                const int w0 = xs >> 5, sh = xs & 31;
                const int wl = p.Ww - 1;
                unsigned long long lo = (unsigned long long)(er[w0] & mr[w0]);
                unsigned long long mid = w0 + 1 <= wl ? (unsigned long long)(er[w0 + 1] & mr[w0 + 1]) : 0ull;
                unsigned long long hi = w0 + 2 <= wl ? (unsigned long long)(er[w0 + 2] & mr[w0 + 2]) : 0ull;
                bits = ((lo | (mid << 32)) >> sh) | (sh ? (hi << (64 - sh)) : 0ull);
            } else {
                for (int t = 0; t < rw && t < 64; ++t) {
                    const int gx = reflect101(xs + t, p.W);
                    const uint32_t w = er[gx >> 5] & mr[gx >> 5];
                    bits |= (unsigned long long)((w >> (gx & 31)) & 1u) << t;
                }
            }
            if (rw < 64) bits &= (1ull << rw) - 1ull;
            rowbits[ty] = bits;
        }
        __syncthreads();
        const int ox_n = min(GT, p.Ws - X0), oy_n = min(GT, p.Hs - Y0);
        // row filter (table lookup; general path for wide kernels or windows > 64 bits)
        const bool small = rh * nsx < 4096 && nsx < 128;      // the magic division holds
        const uint32_t m_nsx = (1u << 19) / (uint32_t)nsx + 1u, m_nox = (1u << 19) / (uint32_t)nox + 1u;
        if (gray) {
            // cv::RowFilter<uchar, double>: s = k[0] * S[0]; s += k[j] * S[j], the u8 samples converted to double
            for (int idx = threadIdx.x; idx < rh * nsx; idx += LG_T) {
                const int ry = idx / nsx, cx = idx - ry * nsx;
                const uint8_t* row = gimg + (size_t)reflect101(sy_lo - h + ry, p.Hc) * p.W;
                double s = 0.0;
                for (int j = 0; j < p.ntaps; ++j) {
                    const double term = p.k[j] * (double)row[reflect101(xs + cx + j, p.W)];
                    if (j == 0) s = term; else s += term;
                }
                F[idx] = s;
            }
        } else if (use_table && rw <= 64) {
            const int msk = (1 << p.ntaps) - 1;
            for (int idx = threadIdx.x; idx < (LF_GRAD_DIAG & 8 ? 0 : rh * nsx); idx += LG_T) {
                const int ry = small ? div_small(idx, m_nsx) : idx / nsx, cx = idx - ry * nsx;
                const unsigned long long rb = rowbits[ry];
                F[idx] = rb ? T[(int)(rb >> cx) & msk] : 0.0;
            }
        } else {
            for (int idx = threadIdx.x; idx < rh * nsx; idx += LG_T) {
                int ry = idx / nsx, cx = idx - ry * nsx;
                const int gy = reflect101(sy_lo - h + ry, p.Hc);
                double s = 0.0;
                for (int j = 0; j < p.ntaps; ++j) {
                    const int gx = reflect101(xs + cx + j, p.W);
                    const uint32_t w = eb[(size_t)gy * p.Ww + (gx >> 5)] & mk[(size_t)gy * p.Ww + (gx >> 5)];
                    const double term = p.k[j] * (((w >> (gx & 31)) & 1u) ? 255.0 : 0.0);
                    if (j == 0) s = term; else s += term;
                }
                F[idx] = s;
            }
        }
        __syncthreads();
        // column filter
        for (int idx = threadIdx.x; idx < (LF_GRAD_DIAG & 1 ? 0 : nsy * nsx); idx += LG_T) {
            const double* S = F + idx + h * nsx;                  // (by + h, cx) of idx = by * nsx + cx
            // F holds sums of positive constants or +0.0, so an all-zero window gives exactly +0.0 through the same
            // arithmetic (testing the window for zero first cost twice the instructions of the seven multiply-adds)
            double s = p.k[h] * S[0] + 0.0;
            for (int j = 1; j <= h; ++j) s += p.k[h + j] * (S[j * nsx] + S[-j * nsx]);
            Bl[idx] = s;
        }
        __syncthreads();
        // horizontal resize
        for (int idx = threadIdx.x; idx < (LF_GRAD_DIAG & 2 ? 0 : nsy * nox); idx += LG_T) {
            const int by = small ? div_small(idx, m_nox) : idx / nox, ox = idx - by * nox;
            const int dx = X0 + ox;
            const int sx = t_xofs[ox];
            const double* S = Bl + by * nsx;
            double v;
            if (dx < rt.xmax) v = S[sx] * (double)t_xa[2 * ox] + S[sx + 1] * (double)t_xa[2 * ox + 1];
            else v = S[sx] * 1.0;
            Hb[by * (GT + 1) + ox] = v;
        }
        __syncthreads();
        // vertical resize
        for (int idx = threadIdx.x; idx < (LF_GRAD_DIAG & 4 ? 0 : noy * nox); idx += LG_T) {
            const int oy = small ? div_small(idx, m_nox) : idx / nox, ox = idx - oy * nox;
            const int r0 = t_y0[oy], r1 = t_y1[oy];
            Sc[oy * (GT + 1) + ox] = Hb[r0 * (GT + 1) + ox] * (double)t_yb[2 * oy] + Hb[r1 * (GT + 1) + ox] * (double)t_yb[2 * oy + 1];
        }
        __syncthreads();
        // gradient + level-line angle; defined pixels are queued for the trigonometry pass
        double local_max = -1.0;
        for (int oy = threadIdx.x >> 5; oy < oy_n; oy += LG_T / 32) {
            const int ox = threadIdx.x & 31;
            if (ox >= ox_n) continue;
            int dx = X0 + ox, dy = Y0 + oy;
            size_t a = (size_t)dy * p.Ws + dx;
            double norm = 0.0;
            if (dx < p.Ws - 1 && dy < p.Hs - 1) {
                const double* q = Sc + oy * (GT + 1) + ox;
                double DA = q[GT + 2] - q[0];
                double BC = q[1] - q[GT + 1];
                double gx = DA + BC, gy = DA - BC;
                const double n2 = (gx * gx + gy * gy) / 4;
                if (n2 != 0.0) {                                   // sqrt(+0) = +0: flat pixels skip the root
                    norm = dm::dsqrt(n2);
                    if (!(norm <= p.rho)) {
                        // the level-line angle waits for the trigonometry pass: there every lane has a defined pixel (here one
                        // pixel in six does, and the arc tangent would run for the whole wave)
                        if (norm > local_max) local_max = norm;
                        const int slot = atomicAdd(&n_def, 1);
                        dl[slot] = make_uint2((uint32_t)a, (uint32_t)(oy * (GT + 1) + ox));
                        dln[slot] = norm;
                    } else if (l_addr) {
                        const int slot = atomicAdd(&n_lo, 1);
                        lla[slot] = (uint32_t)a;
                        lln[slot] = norm;
                    }
                }
            }
        }
        // one global atomic per tile: positive doubles order like their bit patterns
        {
            unsigned long long m = local_max > 0.0 ? (unsigned long long)__double_as_longlong(local_max) : 0ull;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { unsigned long long o = __shfl_xor(m, d); m = o > m ? o : m; }
            if ((threadIdx.x & 63) == 0 && m) atomicMax(&tile_max, m);
        }
        __syncthreads();
        const int nd = n_def;
        if (threadIdx.x == 0) {
            if (tile_max) atomicMax(maxgrad + pc, tile_max);
            rec_base = nd ? atomicAdd(n_rec + pc, nd) : 0;        // reserve this tile's slots in the problem's record list
            low_base = (l_addr && n_lo) ? atomicAdd(n_low + pc, n_lo) : 0;
        }
        __syncthreads();
        // one record per defined pixel; cos/sin of the float-rounded angle (what region growing
        // accumulates) evaluated on full waves.  Record order is arbitrary (k_lsd_order sorts by address).
        // (a problem with more records than the handle's lists hold: nothing is written, the need is reported, the host grows the
        // lists and runs the batch again -- lanefront_api.hip: lsd_records_retry)
        const bool rec_fit = rec_base + nd <= p.rec_cap, low_fit = low_base + n_lo <= p.rec_cap;
        if (threadIdx.x == 0 && rec_need) {
            if (!rec_fit) atomicMax(rec_need, rec_base + nd);
            if (l_addr && !low_fit) atomicMax(rec_need, low_base + n_lo);
        }
        const size_t rb = (size_t)pc * p.rec_cap + rec_base;
        for (int e = threadIdx.x; e < nd && rec_fit; e += LG_T) {
            const uint2 it = dl[e];
            const double* q = Sc + it.y;                          // the pixel's 2x2 neighbourhood again: the same gx, gy as above
            const double DA = q[GT + 2] - q[0];
            const double BC = q[1] - q[GT + 1];
            const double gx = DA + BC, gy = DA - BC;
            const float av = dm::fast_atan2_deg((float)gx, (float)(-gy));
            const double arad = (double)av * DEG_TO_RADS;
            double s_, c_;
            dm::dsincos((double)(float)arad, s_, c_);
            r_addr[rb + e] = it.x;
            r_deg[rb + e] = av;
            r_mod[rb + e] = dln[e];
            r_cs[rb + e] = c_;
            r_sn[rb + e] = s_;
            if (p.r_sd) {                                         // the sums of a region that starts here (LsdParams::r_sd)
                double s2, c2;
                dm::dsincos(arad, s2, c2);
                *reinterpret_cast<float2*>(p.r_sd + 2 * (rb + e)) = make_float2((float)c2, (float)s2);
            }
        }
        if (l_addr) {
            const size_t lb = (size_t)pc * p.rec_cap + low_base;
            for (int e = threadIdx.x; e < n_lo && low_fit; e += LG_T) { l_addr[lb + e] = lla[e]; l_mod[lb + e] = lln[e]; }
        }
    }
}

void launch_lsd_grad(const LsdParams& p, const ResizeTables& rt, int n_frames, const uint32_t* edge_bits,
                     const uint32_t* mask_bits, uint32_t* r_addr, float* r_deg, double* r_mod, double* r_cs, double* r_sn,
                     int* n_rec, unsigned long long* maxgrad, int max_nsx, int max_nsy, uint32_t* list, int* list_count,
                     uint32_t* l_addr, double* l_mod, int* n_low, int* rec_need, bool counters_zeroed, hipStream_t s)
{
    const int h = p.half;
    const size_t szF = (size_t)(max_nsy + 2 * h) * max_nsx, szBl = (size_t)max_nsy * max_nsx;
    const size_t szHb = (size_t)max_nsy * (GT + 1), szSc = (size_t)(GT + 1) * (GT + 1);
    size_t regA = szF > szHb ? szF : szHb;
    const size_t min_a = (size_t)(l_addr ? 4 : 2) * GT * GT;
    if (regA < min_a) regA = min_a;
    const size_t regB = szBl > szSc ? szBl : szSc;
    const size_t lds = sizeof(double) * (regA + regB);
    dim3 grid((p.Hs + GT - 1) / GT, n_frames * 3);
    if (!counters_zeroed) {                                  // (the batch path zeroes all of a batch's counters with one memset)
        (void)hipMemsetAsync(list_count, 0, sizeof(int), s);
        (void)hipMemsetAsync(n_rec, 0, (size_t)n_frames * 3 * sizeof(int), s);
        if (n_low) (void)hipMemsetAsync(n_low, 0, (size_t)n_frames * 3 * sizeof(int), s);
    }
    hipLaunchKernelGGL(k_lsd_classify, grid, dim3(256), 0, s, p, rt, edge_bits, mask_bits, list, list_count);
    const int per_cu = (int)((150 * 1024) / (lds + 3072));
    const int blocks = 256 * (per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu));
    hipLaunchKernelGGL(k_lsd_grad, dim3(blocks), dim3(LG_T), lds, s, p, rt, edge_bits, mask_bits, r_addr, r_deg, r_mod, r_cs,
                       r_sn, n_rec, maxgrad, max_nsx, max_nsy, list, list_count, l_addr, l_mod, n_low, nullptr, rec_need);
}

// every tile of colour 0 of every frame: the tile list of a gray image (nothing to classify)
__global__ void k_lsd_list_all(int n_frames, int ntx, int nty, uint32_t* __restrict__ list, int* __restrict__ list_count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = ntx * nty;
    if (i >= n_frames * per) return;
    const int f = i / per, t = i - f * per;
    list[i] = ((uint32_t)(f * 3) << 16) | ((uint32_t)(t / ntx) << 8) | (uint32_t)(t % ntx);
    if (i == 0) *list_count = n_frames * per;
}

// LSD front half on GRAY images [n_frames][Hc][W] (problem f * 3 holds frame f; problems f * 3 + 1, f * 3 + 2 stay empty)
void launch_lsd_grad_gray(const LsdParams& p, const ResizeTables& rt, int n_frames, const uint8_t* gray, uint32_t* r_addr, float* r_deg,
                          double* r_mod, double* r_cs, double* r_sn, int* n_rec, unsigned long long* maxgrad, int max_nsx, int max_nsy,
                          uint32_t* list, int* list_count, uint32_t* l_addr, double* l_mod, int* n_low, hipStream_t s)
{
    const int h = p.half;
    const size_t szF = (size_t)(max_nsy + 2 * h) * max_nsx, szBl = (size_t)max_nsy * max_nsx;
    const size_t szHb = (size_t)max_nsy * (GT + 1), szSc = (size_t)(GT + 1) * (GT + 1);
    size_t regA = szF > szHb ? szF : szHb;
    const size_t min_a = (size_t)(l_addr ? 4 : 2) * GT * GT;
    if (regA < min_a) regA = min_a;
    const size_t regB = szBl > szSc ? szBl : szSc;
    const size_t lds = sizeof(double) * (regA + regB);
    const int ntx = (p.Ws + GT - 1) / GT, nty = (p.Hs + GT - 1) / GT;
    (void)hipMemsetAsync(n_rec, 0, (size_t)n_frames * 3 * sizeof(int), s);
    if (n_low) (void)hipMemsetAsync(n_low, 0, (size_t)n_frames * 3 * sizeof(int), s);
    hipLaunchKernelGGL(k_lsd_list_all, dim3((n_frames * ntx * nty + 255) / 256), dim3(256), 0, s, n_frames, ntx, nty, list, list_count);
    const int per_cu = (int)((150 * 1024) / (lds + 3072));
    const int blocks = 256 * (per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu));
    hipLaunchKernelGGL(k_lsd_grad, dim3(blocks), dim3(LG_T), lds, s, p, rt, nullptr, nullptr, r_addr, r_deg, r_mod, r_cs,
                       r_sn, n_rec, maxgrad, max_nsx, max_nsy, list, list_count, l_addr, l_mod, n_low, gray, nullptr);
}

}  // namespace lf
