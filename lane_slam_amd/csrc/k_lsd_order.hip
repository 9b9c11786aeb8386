// K_lsd_order: raster ordering + LSD pseudo-ordering of the defined pixels (a-4), one workgroup per
// (frame, colour).
//
// Input: the problem's unordered RECORD list from k_lsd_grad (address, angle, magnitude, cos, sin of
// every pixel whose gradient is defined; tiles append in arbitrary order).
// Output, all in COMPACT index space e = raster rank among the problem's defined pixels:
//   c_xy / c_deg / c_mod / c_cs / c_sn[e]   the record fields gathered in raster order
//   row_start[y]                            first entry of every scaled-image row
//   order[i]                                seeds, (n_bins-1-bin) << 20 | e, sorted: OpenCV lsd.cpp
//       ll_angle (restated, see oracle/lf_oracle_lsd.c lfo_lsd_ll_angle) visits pixels by descending
//       bin = int(modgrad * (n_bins-1)/max_grad), raster order inside a bin; pixels with NOTDEF angle
//       can never seed a region, so only defined pixels are listed.
// k_lsd_grow works entirely in that index space (tens of KB per problem, cache resident).
//
// Integer work, exact and deterministic: (1) LSD radix sort of (address << 32 | record) on the
// 20-bit address, five 4-bit passes -- addresses are unique, so the result does not depend on the
// order in which tiles appended their records; (2) gather + row starts + bins; (3) stable LSD radix
// sort of the seeds on the 10-bit bin key, three 4-bit passes.  Each lane owns a contiguous run of
// items, per-lane bucket counters live in a [16][512] LDS matrix and one wave scans each bucket
// row, so no atomics are needed.
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

template <typename T>
__device__ void radix_pass(const T* __restrict__ src, T* __restrict__ dst, int n, int shift,
                           uint32_t* cnt /*[NB][OT]*/, int* tot /*[NB]*/, int* base /*[NB]*/)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int seg = (n + OT - 1) / OT;
    const int i0 = min(n, t * seg), i1 = min(n, i0 + seg);
    for (int b = 0; b < NB; ++b) cnt[b * OT + t] = 0;
    for (int i = i0; i < i1; ++i) cnt[(int)((src[i] >> shift) & (NB - 1)) * OT + t]++;
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
        const T it = src[i];
        int b = (int)((it >> shift) & (NB - 1));
        uint32_t pos = (uint32_t)base[b] + cnt[b * OT + t]++;
        dst[pos] = it;
    }
    __syncthreads();
}

__global__ __launch_bounds__(OT) void k_lsd_order(LsdParams p, const uint32_t* __restrict__ r_addr,
                                                  const float* __restrict__ r_deg, const double* __restrict__ r_mod,
                                                  const double* __restrict__ r_cs, const double* __restrict__ r_sn,
                                                  const int* __restrict__ n_rec,
                                                  const unsigned long long* __restrict__ maxgrad,
                                                  unsigned long long* __restrict__ sort_a, unsigned long long* __restrict__ sort_b,
                                                  uint32_t* __restrict__ order_a, uint32_t* __restrict__ order_b,
                                                  int* __restrict__ norder, uint32_t* __restrict__ c_xy,
                                                  float* __restrict__ c_deg, double* __restrict__ c_mod,
                                                  double* __restrict__ c_cs, double* __restrict__ c_sn,
                                                  int* __restrict__ row_start)
{
    __shared__ uint32_t cnt[NB * OT];
    __shared__ int tot[NB];
    __shared__ int base[NB];
    const int pc = blockIdx.x;
    const size_t Ps = (size_t)p.Hs * p.Ws;
    const size_t o = (size_t)pc * Ps;
    const int n = n_rec[pc];
    const int t = threadIdx.x;
    unsigned long long* X = sort_a + o;
    unsigned long long* Y = sort_b + o;
    uint32_t* A = order_a + o;
    uint32_t* B = order_b + o;
    int* RS = row_start + (size_t)pc * (p.Hs + 1);
    if (t == 0) norder[pc] = n;
    if (n == 0) {
        for (int y = t; y <= p.Hs; y += OT) RS[y] = 0;
        return;
    }
    // (1) raster order: sort (address, record index) by address
    for (int i = t; i < n; i += OT) X[i] = ((unsigned long long)r_addr[o + i] << 32) | (unsigned int)i;
    __syncthreads();
    radix_pass(X, Y, n, 32, cnt, tot, base);
    radix_pass(Y, X, n, 36, cnt, tot, base);
    radix_pass(X, Y, n, 40, cnt, tot, base);
    radix_pass(Y, X, n, 44, cnt, tot, base);
    radix_pass(X, Y, n, 48, cnt, tot, base);                 // sorted by address in Y
    // (2) gather into the compact arrays, row starts, seed items
    const double max_grad = __longlong_as_double((long long)maxgrad[pc]);
    const double bin_coef = (max_grad > 0) ? (double)(p.n_bins - 1) / max_grad : 0;
    for (int e = t; e < n; e += OT) {
        const unsigned long long it = Y[e];
        const uint32_t addr = (uint32_t)(it >> 32), ri = (uint32_t)it;
        const int y = (int)(addr / (uint32_t)p.Ws), x = (int)(addr - (uint32_t)y * (uint32_t)p.Ws);
        const double m = r_mod[o + ri];
        c_xy[o + e] = ((uint32_t)y << 16) | (uint32_t)x;
        c_deg[o + e] = r_deg[o + ri];
        c_mod[o + e] = m;
        c_cs[o + e] = r_cs[o + ri];
        c_sn[o + e] = r_sn[o + ri];
        const int bin = (int)(m * bin_coef);
        B[e] = ((uint32_t)((p.n_bins - 1) - bin) << 20) | (uint32_t)e;     // seeds carry the compact index
        // rows (y_prev, y] start at e; the first entry also covers rows 0..y
        const int yp = e == 0 ? -1 : (int)((uint32_t)(Y[e - 1] >> 32) / (uint32_t)p.Ws);
        for (int yy = yp + 1; yy <= y; ++yy) RS[yy] = e;
        if (e == n - 1)
            for (int yy = y + 1; yy <= p.Hs; ++yy) RS[yy] = n;
    }
    __syncthreads();
    // (3) pseudo-ordering of the seeds: stable sort on the bin key
    radix_pass(B, A, n, 20, cnt, tot, base);
    radix_pass(A, B, n, 24, cnt, tot, base);
    radix_pass(B, A, n, 28, cnt, tot, base);                 // sorted list ends in order_a
}

void launch_lsd_order(const LsdParams& p, int n_frames, const uint32_t* r_addr, const float* r_deg, const double* r_mod,
                      const double* r_cs, const double* r_sn, const int* n_rec, const unsigned long long* maxgrad,
                      unsigned long long* sort_a, unsigned long long* sort_b, uint32_t* order_a, uint32_t* order_b,
                      int* norder, uint32_t* c_xy, float* c_deg, double* c_mod, double* c_cs, double* c_sn,
                      int* row_start, hipStream_t s)
{
    hipLaunchKernelGGL(k_lsd_order, dim3(n_frames * 3), dim3(OT), 0, s, p, r_addr, r_deg, r_mod, r_cs, r_sn, n_rec, maxgrad,
                       sort_a, sort_b, order_a, order_b, norder, c_xy, c_deg, c_mod, c_cs, c_sn, row_start);
}

// Debug only: dense angle / magnitude planes rebuilt from the compact arrays (NOTDEF / 0 elsewhere).
__global__ void k_lsd_dense_debug(LsdParams p, const int* __restrict__ norder, const uint32_t* __restrict__ c_xy,
                                  const float* __restrict__ c_deg, const double* __restrict__ c_mod,
                                  float* __restrict__ ang, double* __restrict__ mod)
{
    const int pc = blockIdx.x;
    const size_t Ps = (size_t)p.Hs * p.Ws, o = (size_t)pc * Ps;
    for (size_t i = threadIdx.x; i < Ps; i += blockDim.x) { ang[o + i] = kNotDef; mod[o + i] = 0.0; }
    __syncthreads();
    const int n = norder[pc];
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const uint32_t xy = c_xy[o + e];
        const size_t a = (size_t)(xy >> 16) * p.Ws + (xy & 0xffffu);
        ang[o + a] = c_deg[o + e];
        mod[o + a] = c_mod[o + e];
    }
}

void launch_lsd_dense_debug(const LsdParams& p, int n_frames, const int* norder, const uint32_t* c_xy, const float* c_deg,
                            const double* c_mod, float* ang, double* mod, hipStream_t s)
{
    hipLaunchKernelGGL(k_lsd_dense_debug, dim3(n_frames * 3), dim3(1024), 0, s, p, norder, c_xy, c_deg, c_mod, ang, mod);
}

}  // namespace lf
