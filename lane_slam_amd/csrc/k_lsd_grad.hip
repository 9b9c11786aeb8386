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
// One workgroup produces a 32x32 tile of the scaled image.  All intermediates (raw tile with
// blur halo, row-filtered, blurred, h-resized, v-resized) live in LDS; HBM sees one read of
// the bit-packed edges + the u8 mask and one write of angle (f32 degrees), modgrad (f64) and,
// only where the gradient is defined, cos/sin of the float-rounded angle (f64) that region
// growing accumulates.  Tiles whose raw footprint holds no edge pixel skip all arithmetic.
// Algorithmic bytes per scaled pixel: (1/0.64)*(1+1/8) read + 4 + 8 written.
#include "common.h"

namespace lf {

constexpr int GT = 32;   // scaled tile edge

__device__ __forceinline__ int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) { p = p < 0 ? -p : 2 * (n - 1) - p; }
    return p;
}

__global__ __launch_bounds__(256) void k_lsd_grad(LsdParams p, ResizeTables rt, const uint32_t* __restrict__ edge_bits,
                                                  const uint8_t* __restrict__ masks, float* __restrict__ ang,
                                                  double* __restrict__ mod, double* __restrict__ cs,
                                                  double* __restrict__ sn, float2* __restrict__ seedcs,
                                                  unsigned long long* __restrict__ maxgrad, int max_nsx, int max_nsy)
{
    extern __shared__ double lds_d[];
    const int h = p.half;
    const int pc = blockIdx.z;                 // problem = frame*3 + colour
    const int f = pc / 3;
    const int X0 = blockIdx.x * GT, Y0 = blockIdx.y * GT;
    const int X1 = min(X0 + GT, p.Ws - 1), Y1 = min(Y0 + GT, p.Hs - 1);
    const int sx_lo = rt.xofs[X0], sx_hi = min(rt.xofs[X1] + 1, p.W - 1);
    const int sy_lo = rt.y0[Y0], sy_hi = rt.y1[Y1];
    const int nsx = sx_hi - sx_lo + 1, nsy = sy_hi - sy_lo + 1;
    const int nox = X1 - X0 + 1, noy = Y1 - Y0 + 1;   // scaled samples needed (incl. +1 neighbour)
    const int rw = nsx + 2 * h, rh = nsy + 2 * h;

    // LDS carve (doubles first for alignment)
    double* F = lds_d;                                   // [rh][nsx]
    double* Bl = F + (size_t)(max_nsy + 2 * h) * max_nsx;        // [nsy][nsx]
    double* Hb = Bl + (size_t)max_nsy * max_nsx;                 // [nsy][GT+1]
    double* Sc = Hb + (size_t)max_nsy * (GT + 1);                // [GT+1][GT+1]
    uint8_t* raw = reinterpret_cast<uint8_t*>(Sc + (GT + 1) * (GT + 1));   // [rh][rw]

    const size_t P = (size_t)p.Hc * p.W;
    const uint8_t* mk = masks + (size_t)pc * P;
    const uint32_t* eb = edge_bits + (size_t)f * p.Hc * p.Ww;
    int any = 0;
    for (int idx = threadIdx.x; idx < rw * rh; idx += 256) {
        int ty = idx / rw, tx = idx - ty * rw;
        int gx = reflect101(sx_lo - h + tx, p.W), gy = reflect101(sy_lo - h + ty, p.Hc);
        uint32_t w = eb[(size_t)gy * p.Ww + (gx >> 5)];
        uint8_t v = (((w >> (gx & 31)) & 1u) && mk[(size_t)gy * p.W + gx]) ? 255 : 0;
        raw[idx] = v;
        any |= v;
    }
    any = __syncthreads_or(any);

    const size_t Ps = (size_t)p.Hs * p.Ws;
    float* o_ang = ang + (size_t)pc * Ps;
    double* o_mod = mod + (size_t)pc * Ps;
    const int ox_n = min(GT, p.Ws - X0), oy_n = min(GT, p.Hs - Y0);
    if (!any) {
        for (int idx = threadIdx.x; idx < ox_n * oy_n; idx += 256) {
            int oy = idx / ox_n, ox = idx - oy * ox_n;
            size_t a = (size_t)(Y0 + oy) * p.Ws + X0 + ox;
            o_ang[a] = kNotDef;
            o_mod[a] = 0.0;
        }
        return;
    }
    // row filter
    for (int idx = threadIdx.x; idx < rh * nsx; idx += 256) {
        int ry = idx / nsx, cx = idx - ry * nsx;
        const uint8_t* S = raw + ry * rw + cx;
        double s = p.k[0] * (double)S[0];
        for (int j = 1; j < p.ntaps; ++j) s += p.k[j] * (double)S[j];
        F[ry * nsx + cx] = s;
    }
    __syncthreads();
    // column filter
    for (int idx = threadIdx.x; idx < nsy * nsx; idx += 256) {
        int by = idx / nsx, cx = idx - by * nsx;
        const double* S = F + (by + h) * nsx + cx;
        double s = p.k[h] * S[0] + 0.0;
        for (int j = 1; j <= h; ++j) s += p.k[h + j] * (S[j * nsx] + S[-j * nsx]);
        Bl[by * nsx + cx] = s;
    }
    __syncthreads();
    // horizontal resize
    for (int idx = threadIdx.x; idx < nsy * nox; idx += 256) {
        int by = idx / nox, ox = idx - by * nox;
        int dx = X0 + ox;
        int sx = rt.xofs[dx] - sx_lo;
        const double* S = Bl + by * nsx;
        double v;
        if (dx < rt.xmax) v = S[sx] * (double)rt.xa[2 * dx] + S[sx + 1] * (double)rt.xa[2 * dx + 1];
        else v = S[sx] * 1.0;
        Hb[by * (GT + 1) + ox] = v;
    }
    __syncthreads();
    // vertical resize
    for (int idx = threadIdx.x; idx < noy * nox; idx += 256) {
        int oy = idx / nox, ox = idx - oy * nox;
        int dy = Y0 + oy;
        int r0 = rt.y0[dy] - sy_lo, r1 = rt.y1[dy] - sy_lo;
        Sc[oy * (GT + 1) + ox] = Hb[r0 * (GT + 1) + ox] * (double)rt.yb[2 * dy] + Hb[r1 * (GT + 1) + ox] * (double)rt.yb[2 * dy + 1];
    }
    __syncthreads();
    // gradient + angle
    const double DEG_TO_RADS = 3.14159265358979323846 / 180;
    double local_max = -1.0;
    double* o_cs = cs + (size_t)pc * Ps;
    double* o_sn = sn + (size_t)pc * Ps;
    float2* o_seed = seedcs + (size_t)pc * Ps;
    for (int idx = threadIdx.x; idx < ox_n * oy_n; idx += 256) {
        int oy = idx / ox_n, ox = idx - oy * ox_n;
        int dx = X0 + ox, dy = Y0 + oy;
        size_t a = (size_t)dy * p.Ws + dx;
        float av = kNotDef;
        double norm = 0.0;
        if (dx < p.Ws - 1 && dy < p.Hs - 1) {
            const double* q = Sc + oy * (GT + 1) + ox;
            double DA = q[GT + 2] - q[0];
            double BC = q[1] - q[GT + 1];
            double gx = DA + BC, gy = DA - BC;
            norm = dm::dsqrt((gx * gx + gy * gy) / 4);
            if (!(norm <= p.rho)) {
                av = dm::fast_atan2_deg((float)gx, (float)(-gy));
                double arad = (double)av * DEG_TO_RADS;
                double s_, c_;
                dm::dsincos((double)(float)arad, s_, c_);
                o_cs[a] = c_;
                o_sn[a] = s_;
                // a region's first pixel enters the angle sums as float(cos/sin) of the UNROUNDED angle
                dm::dsincos(arad, s_, c_);
                o_seed[a] = make_float2((float)c_, (float)s_);
                if (norm > local_max) local_max = norm;
            }
        }
        o_ang[a] = av;
        o_mod[a] = norm;
    }
    if (local_max > 0.0) atomicMax(maxgrad + pc, (unsigned long long)__double_as_longlong(local_max));
}

void launch_lsd_grad(const LsdParams& p, const ResizeTables& rt, int n_frames, const uint32_t* edge_bits,
                     const uint8_t* masks, float* ang, double* mod, double* cs, double* sn, float2* seedcs,
                     unsigned long long* maxgrad, int max_nsx, int max_nsy, hipStream_t s)
{
    const int h = p.half;
    size_t lds = sizeof(double) * ((size_t)(max_nsy + 2 * h) * max_nsx + (size_t)max_nsy * max_nsx +
                                   (size_t)max_nsy * (GT + 1) + (size_t)(GT + 1) * (GT + 1)) +
                 (size_t)(max_nsy + 2 * h) * (max_nsx + 2 * h);
    dim3 grid((p.Ws + GT - 1) / GT, (p.Hs + GT - 1) / GT, n_frames * 3);
    hipLaunchKernelGGL(k_lsd_grad, grid, dim3(256), lds, s, p, rt, edge_bits, masks, ang, mod, cs, sn, seedcs,
                       maxgrad, max_nsx, max_nsy);
}

}  // namespace lf
