// K_lsd_order: LSD pseudo-ordering of seed pixels (a-4), one workgroup per (frame, colour).
//
// OpenCV lsd.cpp ll_angle (restated, see oracle/lf_oracle_lsd.c lfo_lsd_ll_angle): every
// pixel gets bin = int(modgrad * (n_bins-1)/max_grad); seeds are visited by descending
// bin, raster order inside a bin.  Pixels whose angle is NOTDEF can never seed a region,
// so only defined pixels are listed.
//
// Integer work, exact: (1) stable raster-order compaction of defined pixels into
// items = (1023-bin) << 20 | compact index, (2) a stable LSD radix sort on the 10-bit key, three
// 4-bit passes.  Each lane owns a contiguous run of items, per-lane bucket counters live in
// a [16][512] LDS matrix, and one wave scans each bucket row, so no atomics are needed and
// the order is deterministic.  The compaction also gathers every defined pixel's coordinates,
// angle, magnitude and cos/sin into COMPACT arrays (raster order) plus the first entry of every
// row: k_lsd_grow works entirely in that index space (a few hundred KB per problem, cache
// resident) and never touches the 96 %-undefined dense planes.
#include "common.h"

namespace lf {

constexpr int OT = 512;          // threads
constexpr int NB = 16;           // buckets per pass (4-bit digits; [16][512] u32 = 32 KB LDS)

__device__ __forceinline__ int wave_incl_scan(int v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int n = __shfl_up(v, d);
        if (lane >= d) v += n;
    }
    return v;
}

__device__ void radix_pass(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int n, int shift,
                           uint32_t* cnt /*[NB][OT]*/, int* tot /*[NB]*/, int* base /*[NB]*/)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int seg = (n + OT - 1) / OT;
    const int i0 = min(n, t * seg), i1 = min(n, i0 + seg);
    for (int b = 0; b < NB; ++b) cnt[b * OT + t] = 0;
    for (int i = i0; i < i1; ++i) cnt[((src[i] >> shift) & (NB - 1)) * OT + t]++;
    __syncthreads();
    // wave `wave` scans bucket rows wave*2, wave*2+1 (8 waves x 2 = 16 rows)
    for (int bb = 0; bb < NB / (OT / 64); ++bb) {
        const int b = wave * (NB / (OT / 64)) + bb;
        int carry = 0;
        for (int c = 0; c < OT / 64; ++c) {
            int v = (int)cnt[b * OT + c * 64 + lane];
            int inc = wave_incl_scan(v, lane);
            cnt[b * OT + c * 64 + lane] = (uint32_t)(carry + inc - v);
            carry += __shfl(inc, 63);
        }
        if (lane == 0) tot[b] = carry;
    }
    __syncthreads();
    if (wave == 0) {
        int v = lane < NB ? tot[lane] : 0;
        int inc = wave_incl_scan(v, lane);
        if (lane < NB) base[lane] = inc - v;
    }
    __syncthreads();
    for (int i = i0; i < i1; ++i) {
        uint32_t it = src[i];
        int b = (it >> shift) & (NB - 1);
        uint32_t pos = (uint32_t)base[b] + cnt[b * OT + t]++;
        dst[pos] = it;
    }
    __syncthreads();
}

__global__ __launch_bounds__(OT) void k_lsd_order(LsdParams p, const float* __restrict__ ang,
                                                  const double* __restrict__ mod,
                                                  const unsigned long long* __restrict__ maxgrad,
                                                  uint32_t* __restrict__ order_a, uint32_t* __restrict__ order_b,
                                                  int* __restrict__ norder, const double* __restrict__ cs,
                                                  const double* __restrict__ sn, uint32_t* __restrict__ c_xy,
                                                  float* __restrict__ c_deg, double* __restrict__ c_mod,
                                                  double* __restrict__ c_cs, double* __restrict__ c_sn,
                                                  int* __restrict__ row_start)
{
    __shared__ uint32_t cnt[NB * OT];
    __shared__ int tot[NB];
    __shared__ int base[NB];
    __shared__ int wsum[OT / 64];
    const int pc = blockIdx.x;
    const size_t Ps = (size_t)p.Hs * p.Ws;
    const float* a = ang + (size_t)pc * Ps;
    const double* m = mod + (size_t)pc * Ps;
    uint32_t* A = order_a + (size_t)pc * Ps;
    uint32_t* B = order_b + (size_t)pc * Ps;
    // compact per-pixel arrays, indexed by e = raster rank among the problem's defined pixels
    uint32_t* CXY = c_xy + (size_t)pc * Ps;
    float* CDEG = c_deg + (size_t)pc * Ps;
    double* CMOD = c_mod + (size_t)pc * Ps;
    double* CCS = c_cs + (size_t)pc * Ps;
    double* CSN = c_sn + (size_t)pc * Ps;
    const double* csd = cs + (size_t)pc * Ps;
    const double* snd = sn + (size_t)pc * Ps;
    int* RS = row_start + (size_t)pc * (p.Hs + 1);         // first list entry of every scaled-image row
    const double max_grad = __longlong_as_double((long long)maxgrad[pc]);
    const double bin_coef = (max_grad > 0) ? (double)(p.n_bins - 1) / max_grad : 0;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

    int running = 0;
    // stable raster-order compaction, 4 pixels per lane per sweep (one 16-byte load of the angle plane)
    const bool vec = (Ps & 3) == 0;
    const size_t step = vec ? (size_t)OT * 4 : (size_t)OT;
    for (size_t c0 = 0; c0 < Ps; c0 += step) {
        float av[4] = {kNotDef, kNotDef, kNotDef, kNotDef};
        const size_t i0 = vec ? c0 + (size_t)t * 4 : c0 + t;
        if (vec) {
            if (i0 < Ps) { const float4 q = *reinterpret_cast<const float4*>(a + i0); av[0] = q.x; av[1] = q.y; av[2] = q.z; av[3] = q.w; }
        } else if (i0 < Ps) av[0] = a[i0];
        const int npx = vec ? 4 : 1;
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) cnt += (k < npx && av[k] != kNotDef) ? 1 : 0;
        const int inc = wave_incl_scan(cnt, lane);
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int off = running, total = 0;
        for (int w = 0; w < OT / 64; ++w) { int s_ = wsum[w]; if (w < wave) off += s_; total += s_; }
        int e = off + inc - cnt;                           // entries before this lane's first pixel
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t i = i0 + k;
            if (k >= npx || i >= Ps) continue;
            const int y = (int)(i / p.Ws), x = (int)(i - (size_t)y * p.Ws);
            if (x == 0) RS[y] = e;
            if (av[k] != kNotDef) {
                int bin = (int)(m[i] * bin_coef);
                uint32_t key = (uint32_t)((p.n_bins - 1) - bin);
                B[e] = (key << 20) | (uint32_t)e;        // seeds carry the compact index
                CXY[e] = ((uint32_t)y << 16) | (uint32_t)x;
                CDEG[e] = av[k];
                CMOD[e] = m[i];
                CCS[e] = csd[i];
                CSN[e] = snd[i];
                ++e;
            }
        }
        running += total;
        __syncthreads();
    }
    const int n = running;
    if (t == 0) { norder[pc] = n; RS[p.Hs] = n; }
    __syncthreads();
    radix_pass(B, A, n, 20, cnt, tot, base);
    radix_pass(A, B, n, 24, cnt, tot, base);
    radix_pass(B, A, n, 28, cnt, tot, base);     // sorted list ends in order_a
}

void launch_lsd_order(const LsdParams& p, int n_frames, const float* ang, const double* mod,
                      const unsigned long long* maxgrad, uint32_t* order_a, uint32_t* order_b, int* norder,
                      const double* cs, const double* sn, uint32_t* c_xy, float* c_deg, double* c_mod, double* c_cs,
                      double* c_sn, int* row_start, hipStream_t s)
{
    hipLaunchKernelGGL(k_lsd_order, dim3(n_frames * 3), dim3(OT), 0, s, p, ang, mod, maxgrad, order_a, order_b,
                       norder, cs, sn, c_xy, c_deg, c_mod, c_cs, c_sn, row_start);
}

}  // namespace lf
