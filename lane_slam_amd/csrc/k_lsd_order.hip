// K_lsd_order: LSD pseudo-ordering of seed pixels (a-4), one workgroup per (frame, colour).
//
// OpenCV lsd.cpp ll_angle (restated, see oracle/lf_oracle_lsd.c lfo_lsd_ll_angle): every
// pixel gets bin = int(modgrad * (n_bins-1)/max_grad); seeds are visited by descending
// bin, raster order inside a bin.  Pixels whose angle is NOTDEF can never seed a region,
// so only defined pixels are listed.
//
// Integer work, exact: (1) stable raster-order compaction of defined pixels into
// items = (1023-bin) << 20 | address, (2) a stable LSD radix sort on the 10-bit key, three
// 4-bit passes.  Each lane owns a contiguous run of items, per-lane bucket counters live in
// a [16][512] LDS matrix, and one wave scans each bucket row, so no atomics are needed and
// the order is deterministic.  The compaction also leaves, per problem, the raster-ordered list
// of defined pixels with their angles and the first entry of every row: k_lsd_grow's NFA
// rectangle counts walk these row lists instead of the (96 % undefined) angle plane.
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
                                                  int* __restrict__ norder, uint2* __restrict__ deflist,
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
    uint2* DL = deflist + (size_t)pc * Ps;                 // raster-ordered (y<<16|x, angle bits) of defined pixels
    int* RS = row_start + (size_t)pc * (p.Hs + 1);         // first list entry of every scaled-image row
    const double max_grad = __longlong_as_double((long long)maxgrad[pc]);
    const double bin_coef = (max_grad > 0) ? (double)(p.n_bins - 1) / max_grad : 0;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

    int running = 0;
    for (size_t c0 = 0; c0 < Ps; c0 += OT) {
        size_t i = c0 + t;
        bool def = i < Ps && a[i] != kNotDef;
        unsigned long long bal = __ballot(def);
        int pre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int off = running;
        int total = 0;
        for (int w = 0; w < OT / 64; ++w) { int s = wsum[w]; if (w < wave) off += s; total += s; }
        if (i < Ps) {
            const int y = (int)(i / p.Ws), x = (int)(i - (size_t)y * p.Ws);
            if (x == 0) RS[y] = off + pre;
            if (def) {
                int bin = (int)(m[i] * bin_coef);
                uint32_t key = (uint32_t)((p.n_bins - 1) - bin);
                B[off + pre] = (key << 20) | (uint32_t)i;
                DL[off + pre] = make_uint2(((uint32_t)y << 16) | (uint32_t)x, __float_as_uint(a[i]));
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
                      uint2* deflist, int* row_start, hipStream_t s)
{
    hipLaunchKernelGGL(k_lsd_order, dim3(n_frames * 3), dim3(OT), 0, s, p, ang, mod, maxgrad, order_a, order_b,
                       norder, deflist, row_start);
}

}  // namespace lf
