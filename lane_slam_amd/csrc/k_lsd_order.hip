// K_lsd_order: raster ordering + LSD pseudo-ordering of the defined pixels (a-4), one workgroup per
// (frame, colour).
//
// Input: the problem's unordered RECORD list from k_lsd_grad (address, angle, magnitude, cos, sin of
// every pixel whose gradient is defined; tiles append in arbitrary order).
// Output, all in COMPACT index space e = raster rank among the problem's defined pixels:
//   c_xy / c_deg / c_mod / c_cs / c_sn[e]   the record fields gathered in raster order (c_cs, c_sn: ONE array of (cos, sin) pairs --
//                                           c_sn = c_cs + 1, both indexed 2 e: k_lsd_grow fetches a candidate's pair with one load)
//   row_start[y]                            first entry of every scaled-image row
//   order[i]                                seeds, (n_bins-1-bin) << 20 | e, sorted: OpenCV lsd.cpp
//       ll_angle (restated, see oracle/lf_oracle_lsd.c lfo_lsd_ll_angle) visits pixels by descending
//       bin = int(modgrad * (n_bins-1)/max_grad), raster order inside a bin; pixels with NOTDEF angle
//       can never seed a region, so only defined pixels are listed.
// k_lsd_grow works entirely in that index space (tens of KB per problem, cache resident).
//
// Integer work, exact and deterministic.  Problems of up to LDS_ITEMS defined pixels stay in LDS: (1) counting
// sort on the image row with atomic cursors + rank among row mates (addresses are unique, so the result does
// not depend on the order in which tiles appended their records); (2) gather + row starts + bins; (3) two
// stable 5-bit counting passes over the 10-bit bin key, ranks from wave ballots.  Larger problems take the
// same three steps as LSD radix sorts with the items in HBM (five + three 4-bit passes; each lane owns a
// contiguous run of items, per-lane bucket counters live in a [16][512] LDS matrix and one wave scans each
// bucket row, so no atomics are needed).
#include <cstdlib>
#include <algorithm>
#include "common.h"
#include <type_traits>
#include "lsd_bitplane.h"

namespace lf {
using std::max;
using std::min;

// four waves per workgroup for the bit-plane ordering (OBT) and the labelling (LT) (512 threads until the end of round 4): camera frames
// 84.4 k -> 85.7 - 87.1 k frames/s, clutter 44.8 k -> 45.5 k, lane frames equal (same-call A/B).  The sorting kernel of rounds 1 - 3 (OT),
// which the 1080p geometries still use, keeps 512: with 256 it was 35 % slower on their larger problems.
#ifndef LF_ORDER_THREADS
#define LF_ORDER_THREADS 512
#endif
#ifndef LF_ORDER_BM_THREADS
#define LF_ORDER_BM_THREADS 256
#endif
#ifndef LF_LABEL_THREADS
#define LF_LABEL_THREADS 256
#endif
constexpr int OT = LF_ORDER_THREADS;          // threads of k_lsd_order
constexpr int OBT = LF_ORDER_BM_THREADS;      // threads of k_lsd_order_bm
constexpr int NB = 16;           // buckets per pass of k_lsd_order (4-bit digits; [16][512] u32 = 32 KB LDS)
constexpr int LDS_ITEMS = 8192;  // problems up to this many defined pixels are ordered entirely in LDS (2 x 32 KB, dynamic)

__device__ __forceinline__ int wave_incl_scan(int v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int n = __shfl_up(v, d);
        if (lane >= d) v += n;
    }
    return v;
}

template <typename T, int NB = 16, int NT = OT>
__device__ __forceinline__ void radix_pass(const T* __restrict__ src, T* __restrict__ dst, int n, int shift,
                           uint32_t* cnt /*[NB][NT]*/, int* tot /*[NB]*/, int* base /*[NB]*/)
{
    static_assert(NB % (NT / 64) == 0, "every wave scans NB / waves bucket rows");
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int seg = (n + NT - 1) / NT;
    const int i0 = min(n, t * seg), i1 = min(n, i0 + seg);
    for (int b = 0; b < NB; ++b) cnt[b * NT + t] = 0;
    for (int i = i0; i < i1; ++i) cnt[(int)((src[i] >> shift) & (NB - 1)) * NT + t]++;
    __syncthreads();
    // wave `wave` scans bucket rows wave*2, wave*2+1 (8 waves x 2 = 16 rows)
    for (int bb = 0; bb < NB / (NT / 64); ++bb) {
        const int b = wave * (NB / (NT / 64)) + bb;
        int carry = 0;
        for (int c = 0; c < NT / 64; ++c) {
            int v = (int)cnt[b * NT + c * 64 + lane];
            int inc = wave_incl_scan(v, lane);
            cnt[b * NT + c * 64 + lane] = (uint32_t)(carry + inc - v);
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
        uint32_t pos = (uint32_t)base[b] + cnt[b * NT + t]++;
        dst[pos] = it;
    }
    __syncthreads();
}

// One STABLE 5-bit counting pass over n <= LDS_ITEMS items held in LDS, ranks from wave ballots.
// Wave w owns the contiguous run [w * C, (w + 1) * C); in every 64-item step a lane's rank among the
// lanes with the same digit comes from five ballots, and the wave's running per-digit count lives in
// LDS (wrun).  After a barrier one wave turns the [wave][digit] counts into scatter bases.
template <typename Dst>
__device__ __forceinline__ void stable_pass5(const uint32_t* src, Dst* dst, int n, int shift,
                                             uint32_t* wrun /*[OT/64][32]*/, uint32_t* wbase /*[OT/64][32]*/)
{
    constexpr int W = OT / 64, MAXSTEP = LDS_ITEMS / OT;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int C = ((n + W - 1) / W + 63) & ~63;                  // whole 64-item steps per wave
    const int start = w * C, end = min(n, start + C);
    if (lane < 32) wrun[w * 32 + lane] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    uint32_t item[MAXSTEP], lrank[MAXSTEP];
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int st = 0; st < MAXSTEP; ++st) {
        const int i = start + st * 64 + lane;
        const bool valid = st * 64 < C && i < end;
        const uint32_t it = valid ? src[i] : 0u;
        const uint32_t d = (it >> shift) & 31u;
        unsigned long long mask = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            const unsigned long long bal = __ballot((d >> b) & 1u);
            mask &= ((d >> b) & 1u) ? bal : ~bal;
        }
        item[st] = it;
        lrank[st] = 0;
        if (valid) {
            lrank[st] = wrun[w * 32 + d] + (uint32_t)__popcll(mask & lt);
            if ((mask >> lane) >> 1 == 0ull) wrun[w * 32 + d] += (uint32_t)__popcll(mask);    // last lane of the group
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    if (t < 32) {
        uint32_t tot = 0;
        for (int k = 0; k < W; ++k) tot += wrun[k * 32 + t];
        uint32_t inc = (uint32_t)wave_incl_scan((int)tot, t);
        uint32_t run = inc - tot;
        for (int k = 0; k < W; ++k) { wbase[k * 32 + t] = run; run += wrun[k * 32 + t]; }
    }
    __syncthreads();
#pragma unroll
    for (int st = 0; st < MAXSTEP; ++st) {
        const int i = start + st * 64 + lane;
        if (st * 64 < C && i < end) dst[wbase[w * 32 + ((item[st] >> shift) & 31u)] + lrank[st]] = item[st];
    }
    __syncthreads();
}

__global__ __launch_bounds__(OT) void k_lsd_order(LsdParams p, const uint32_t* __restrict__ r_addr,
                                                  const float* __restrict__ r_deg, const double* __restrict__ r_mod,
                                                  const double* __restrict__ r_cs, const double* __restrict__ r_sn,
                                                  int* __restrict__ n_rec,
                                                  const unsigned long long* __restrict__ maxgrad,
                                                  unsigned long long* __restrict__ sort_a, unsigned long long* __restrict__ sort_b,
                                                  uint32_t* __restrict__ order_a, uint32_t* __restrict__ order_b,
                                                  int* __restrict__ norder, uint32_t* __restrict__ c_xy,
                                                  float* __restrict__ c_deg, double* __restrict__ c_mod,
                                                  double* __restrict__ c_cs, double* __restrict__ c_sn,
                                                  int* __restrict__ row_start)
{
    // dynamic LDS: [LA LDS_ITEMS][LB LDS_ITEMS][row tables 2 x (Hs + 2)][wave tables 2 x 256]; the HBM path reuses the
    // front of it as its [16][512] counter matrix
    extern __shared__ uint32_t dyn_lds[];
    uint32_t* LA = dyn_lds;
    uint32_t* LB = dyn_lds + LDS_ITEMS;
    uint32_t* cnt = dyn_lds;
    __shared__ int tot[NB];
    __shared__ int base[NB];
    const int pc = blockIdx.x;
    const size_t Ps = (size_t)p.Hs * p.Ws;
    const size_t o = (size_t)pc * p.rec_cap;
    // (more records than the handle's lists hold: k_lsd_grad wrote none and reported the need -- an empty problem until the host has
    // grown the lists and runs the batch again; later kernels read the count from n_rec too)
    const int n_raw = n_rec[pc];
    const int n = n_raw > p.rec_cap ? 0 : n_raw;
    if (n_raw > p.rec_cap && threadIdx.x == 0) n_rec[pc] = 0;
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
    const int idx_bits = __clz((int)(Ps - 1));      // 32 - (bits of the largest pixel address): room for the record number
    const double max_grad = __longlong_as_double((long long)maxgrad[pc]);
    const double bin_coef = (max_grad > 0) ? (double)(p.n_bins - 1) / max_grad : 0;
    if (n <= LDS_ITEMS && n <= (1 << idx_bits) && p.n_bins <= 1024) {
        // Typical problem (a few thousand defined pixels): two counting sorts in LDS.  Keys are unique
        // (addresses), so the raster sort need not be stable: items are dropped into their image row with an
        // LDS atomic cursor in whatever order they arrive,
        // and each item's final place is the row start plus the number of row mates to its left (rows hold tens of
        // entries).  The seeds then take two stable ballot-ranked passes over the bin key.
        uint32_t* rowS = dyn_lds + 2 * LDS_ITEMS;   // [Hs + 1] first entry of every row (exclusive scan)
        uint32_t* rowC = rowS + (p.Hs + 2);          // [Hs + 1] histogram, then scatter cursor
        const uint32_t idx_mask = (1u << idx_bits) - 1u;
        for (int y = t; y <= p.Hs; y += OT) rowC[y] = 0;
        __syncthreads();
        for (int i = t; i < n; i += OT) {
            const uint32_t addr = r_addr[o + i];
            LA[i] = (addr << idx_bits) | (uint32_t)i;   // address above, record number below
            atomicAdd(&rowC[addr / (uint32_t)p.Ws], 1u);
        }
        __syncthreads();
        if (t < 64) {                                // exclusive scan over the rows, one wave
            uint32_t carry = 0;
            for (int y0 = 0; y0 <= p.Hs; y0 += 64) {
                const int y = y0 + t;
                const uint32_t v = y < p.Hs ? rowC[y] : 0u;
                const uint32_t inc = (uint32_t)wave_incl_scan((int)v, t);
                if (y <= p.Hs) { rowS[y] = carry + inc - v; RS[y] = (int)(carry + inc - v); }
                carry += (uint32_t)__shfl((int)inc, 63);
            }
        }
        __syncthreads();
        for (int y = t; y <= p.Hs; y += OT) rowC[y] = rowS[y];
        __syncthreads();
        for (int i = t; i < n; i += OT) {
            const uint32_t it = LA[i];
            const uint32_t y = (it >> idx_bits) / (uint32_t)p.Ws;
            LB[atomicAdd(&rowC[y], 1u)] = it;
        }
        __syncthreads();
        for (int j = t; j < n; j += OT) {
            const uint32_t it = LB[j];
            const uint32_t addr = it >> idx_bits, ri = it & idx_mask;
            const int y = (int)(addr / (uint32_t)p.Ws), x = (int)(addr - (uint32_t)y * (uint32_t)p.Ws);
            const uint32_t s0 = rowS[y], s1 = rowS[y + 1];
            uint32_t e = s0;
            for (uint32_t k = s0; k < s1; ++k) e += LB[k] < it ? 1u : 0u;                // row mates to the left
            const double m = r_mod[o + ri];
            c_xy[o + e] = ((uint32_t)y << 16) | (uint32_t)x;
            c_deg[o + e] = r_deg[o + ri];
            c_mod[o + e] = m;
            c_cs[2 * (o + e)] = r_cs[o + ri];
            c_sn[2 * (o + e)] = r_sn[o + ri];
            if (p.c_sd) *reinterpret_cast<float2*>(p.c_sd + 2 * (size_t)(o + e)) = *reinterpret_cast<const float2*>(p.r_sd + 2 * (size_t)(o + ri));
            const uint32_t key = (uint32_t)((p.n_bins - 1) - (int)(m * bin_coef));
            LA[e] = (key << 20) | e;
        }
        __syncthreads();
        // seeds: stable by construction (LA is in raster order), two 5-bit passes over the 10-bit bin key.
        // The bins of a binary edge image are few and crowded, so a rank-among-bin-mates loop would be quadratic.
        uint32_t* wrun = rowC + (p.Hs + 2);
        uint32_t* wbase = wrun + (OT / 64) * 32;
        stable_pass5(LA, LB, n, 20, wrun, wbase);
        stable_pass5(LB, A, n, 25, wrun, wbase);                  // k_lsd_grow reads the seeds from order_a
        return;
    }
    // ---- large problems: same algorithm with the items in HBM
    // (1) raster order: sort (address, record index) by address
    for (int i = t; i < n; i += OT) X[i] = ((unsigned long long)r_addr[o + i] << 32) | (unsigned int)i;
    __syncthreads();
    radix_pass(X, Y, n, 32, cnt, tot, base);
    radix_pass(Y, X, n, 36, cnt, tot, base);
    radix_pass(X, Y, n, 40, cnt, tot, base);
    radix_pass(Y, X, n, 44, cnt, tot, base);
    radix_pass(X, Y, n, 48, cnt, tot, base);                 // sorted by address in Y
    // (2) gather into the compact arrays, row starts, seed items
    for (int e = t; e < n; e += OT) {
        const unsigned long long it = Y[e];
        const uint32_t addr = (uint32_t)(it >> 32), ri = (uint32_t)it;
        const int y = (int)(addr / (uint32_t)p.Ws), x = (int)(addr - (uint32_t)y * (uint32_t)p.Ws);
        const double m = r_mod[o + ri];
        c_xy[o + e] = ((uint32_t)y << 16) | (uint32_t)x;
        c_deg[o + e] = r_deg[o + ri];
        c_mod[o + e] = m;
        c_cs[2 * (o + e)] = r_cs[o + ri];
        c_sn[2 * (o + e)] = r_sn[o + ri];
            if (p.c_sd) *reinterpret_cast<float2*>(p.c_sd + 2 * (size_t)(o + e)) = *reinterpret_cast<const float2*>(p.r_sd + 2 * (size_t)(o + ri));
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

// K_lsd_order_bm (round 4): the same outputs without a sort for the raster order.  The defined pixels of a problem are the set
// bits of a bit plane of its scaled image; a pixel's compact index e -- its place among them in raster order -- is the number of
// set bits in front of it (lsd_bitplane.h).  So: (A) build the plane and its running counts in LDS from the unordered records
// (one LDS atomic per record); (B) every record goes straight to its place rank(address) in the compact arrays, row_start[y] =
// rank(y * Ws), and the seed item (bin key << 20 | e) to a list in raster order; (C) one stable counting pass over the 10-bit bin
// key: every wave owns a contiguous piece of that list and walks it 64 items at a time, an item's place among the items of its
// bin = the wave's running count of the bin + the same-bin lanes below it (ten ballots); the walk is done twice, first to count
// per (wave, bin), then -- the counts turned into start positions -- to place.
// LDS: 9 bytes per 64 pixels for (A) and (B), 16 KB of (wave, bin) counters for (C) in the same space: 18.5 KB at 512 x 256
// whatever the number of defined pixels, against the 64 KB of k_lsd_order (whose problems beyond 8192 records sorted in HBM,
// eight passes): a workgroup finds room beside k_lsd_grow's (25 KB each), and the largest problem of a camera frame (24 k
// records) no longer takes 0.8 ms.  Problems it cannot take (more than 65 535 records: u16 counters) sort in HBM as before, with
// 8-bucket passes whose counter matrix fits the same LDS.
constexpr int OB_BINS = 1024;
__global__ __launch_bounds__(OBT) void k_lsd_order_bm(LsdParams p, const uint32_t* __restrict__ r_addr,
                                                     const float* __restrict__ r_deg, const double* __restrict__ r_mod,
                                                     const double* __restrict__ r_cs, const double* __restrict__ r_sn,
                                                     int* __restrict__ n_rec,
                                                     const unsigned long long* __restrict__ maxgrad,
                                                     unsigned long long* __restrict__ sort_a, unsigned long long* __restrict__ sort_b,
                                                     uint32_t* __restrict__ order_a, uint32_t* __restrict__ order_b,
                                                     int* __restrict__ norder, uint32_t* __restrict__ c_xy,
                                                     float* __restrict__ c_deg, double* __restrict__ c_mod,
                                                     double* __restrict__ c_cs, double* __restrict__ c_sn,
                                                     int* __restrict__ row_start, int force_hbm)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn_lds[];
    __shared__ int tot[8];
    __shared__ int base[8];
    __shared__ int wave_tot[OBT / 64];
    __shared__ uint32_t bin_base[OB_BINS];
    const int pc = blockIdx.x;
    const size_t Ps = (size_t)p.Hs * p.Ws;
    const size_t o = (size_t)pc * p.rec_cap;
    // (more records than the handle's lists hold: k_lsd_grad wrote none and reported the need -- an empty problem until the host has
    // grown the lists and runs the batch again; later kernels read the count from n_rec too)
    const int n_raw = n_rec[pc];
    const int n = n_raw > p.rec_cap ? 0 : n_raw;
    if (n_raw > p.rec_cap && threadIdx.x == 0) n_rec[pc] = 0;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    uint32_t* A = order_a + o;
    uint32_t* B = order_b + o;
    int* RS = row_start + (size_t)pc * (p.Hs + 1);
    if (t == 0) norder[pc] = n;
    if (n == 0) {
        for (int y = t; y <= p.Hs; y += OBT) RS[y] = 0;
        return;
    }
    const double max_grad = __longlong_as_double((long long)maxgrad[pc]);
    const double bin_coef = (max_grad > 0) ? (double)(p.n_bins - 1) / max_grad : 0;
    if (n > 65535 || force_hbm) {
        // ---- the items in HBM, 3-bit digits ([8][512] u32 = 16 KB of counters): raster order by address, then the seeds by bin
        unsigned long long* X = sort_a + o;
        unsigned long long* Y = sort_b + o;
        for (int i = t; i < n; i += OBT) X[i] = ((unsigned long long)r_addr[o + i] << 32) | (unsigned int)i;
        __syncthreads();
        for (int k = 0; k < 8; ++k) {                       // 24 address bits
            radix_pass<unsigned long long, 8, OBT>(k & 1 ? Y : X, k & 1 ? X : Y, n, 32 + 3 * k, dyn_lds, tot, base);
        }
        for (int e = t; e < n; e += OBT) {                   // sorted by address in X
            const unsigned long long it = X[e];
            const uint32_t addr = (uint32_t)(it >> 32), ri = (uint32_t)it;
            const int y = (int)(addr / (uint32_t)p.Ws), x = (int)(addr - (uint32_t)y * (uint32_t)p.Ws);
            const double m = r_mod[o + ri];
            c_xy[o + e] = ((uint32_t)y << 16) | (uint32_t)x;
            c_deg[o + e] = r_deg[o + ri];
            c_mod[o + e] = m;
            c_cs[2 * (o + e)] = r_cs[o + ri];
            c_sn[2 * (o + e)] = r_sn[o + ri];
            if (p.c_sd) *reinterpret_cast<float2*>(p.c_sd + 2 * (size_t)(o + e)) = *reinterpret_cast<const float2*>(p.r_sd + 2 * (size_t)(o + ri));
            const int bin = (int)(m * bin_coef);
            A[e] = ((uint32_t)((p.n_bins - 1) - bin) << 20) | (uint32_t)e;
            const int yp = e == 0 ? -1 : (int)((uint32_t)(X[e - 1] >> 32) / (uint32_t)p.Ws);
            for (int yy = yp + 1; yy <= y; ++yy) RS[yy] = e;
            if (e == n - 1)
                for (int yy = y + 1; yy <= p.Hs; ++yy) RS[yy] = n;
        }
        __syncthreads();
        radix_pass<uint32_t, 8, OBT>(A, B, n, 20, dyn_lds, tot, base);
        radix_pass<uint32_t, 8, OBT>(B, A, n, 23, dyn_lds, tot, base);
        radix_pass<uint32_t, 8, OBT>(A, B, n, 26, dyn_lds, tot, base);
        radix_pass<uint32_t, 8, OBT>(B, A, n, 29, dyn_lds, tot, base);             // 12 key bits; the sorted list ends in order_a
        return;
    }
    // (A) the plane
    bitplane_build<OBT, false>(dyn_lds, r_addr + o, n, p.Ws, Ps, wave_tot);
    __syncthreads();
    // (B) records to their places; seed items in raster order into B
    for (int i = t; i < n; i += OBT) {
        const uint32_t addr = r_addr[o + i];
        const uint32_t e = bitplane_rank(dyn_lds, Ps, (int)addr);
        const int y = (int)(addr / (uint32_t)p.Ws), x = (int)(addr - (uint32_t)y * (uint32_t)p.Ws);
        const double m = r_mod[o + i];
        c_xy[o + e] = ((uint32_t)y << 16) | (uint32_t)x;
        c_deg[o + e] = r_deg[o + i];
        c_mod[o + e] = m;
        c_cs[2 * (o + e)] = r_cs[o + i];
        c_sn[2 * (o + e)] = r_sn[o + i];
            if (p.c_sd) *reinterpret_cast<float2*>(p.c_sd + 2 * (size_t)(o + e)) = *reinterpret_cast<const float2*>(p.r_sd + 2 * (size_t)(o + i));
        B[e] = ((uint32_t)((p.n_bins - 1) - (int)(m * bin_coef)) << 20) | e;
    }
    for (int y = t; y <= p.Hs; y += OBT) RS[y] = (int)bitplane_rank(dyn_lds, Ps, y * p.Ws);
    __syncthreads();                                         // B complete (and visible: same workgroup, barrier), the plane free
    // (C) stable counting pass over the bin key
    uint16_t* wcnt = reinterpret_cast<uint16_t*>(dyn_lds);   // [OBT / 64][OB_BINS]
    for (int i = t; i < (OBT / 64) * OB_BINS / 2; i += OBT) dyn_lds[i] = 0u;
    __syncthreads();
    const int C = ((n + OBT / 64 - 1) / (OBT / 64) + 63) & ~63;       // whole 64-item steps per wave
    const int start = w * C, end = min(n, start + C);
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    uint16_t* mine = wcnt + w * OB_BINS;
    auto walk = [&](bool place) {
        for (int i0 = start; i0 < end; i0 += 64) {
            const int i = i0 + lane;
            const bool valid = i < end;
            const uint32_t it = valid ? B[i] : 0u;
            const uint32_t d = it >> 20;
            unsigned long long mask = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 10; ++b) {
                const unsigned long long bal = __ballot((d >> b) & 1u);
                mask &= ((d >> b) & 1u) ? bal : ~bal;
            }
            if (valid) {
                const uint32_t run = mine[d];
                if (place) A[run + (uint32_t)__popcll(mask & lt)] = it;
                if ((mask >> lane) >> 1 == 0ull) mine[d] = (uint16_t)(run + (uint32_t)__popcll(mask));       // last lane of the group
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    };
    walk(false);
    __syncthreads();
    // counts -> start positions: bins ascending, waves ascending inside a bin
    for (int d = t; d < OB_BINS; d += OBT) {
        uint32_t s_ = 0;
        for (int k = 0; k < OBT / 64; ++k) s_ += wcnt[k * OB_BINS + d];
        bin_base[d] = s_;
    }
    __syncthreads();
    if (w == 0) {
        uint32_t carry = 0;
        for (int d0 = 0; d0 < OB_BINS; d0 += 64) {
            const uint32_t v = bin_base[d0 + lane];
            const uint32_t inc = (uint32_t)wave_incl_scan((int)v, lane);
            bin_base[d0 + lane] = carry + inc - v;
            carry += (uint32_t)__shfl((int)inc, 63);
        }
    }
    __syncthreads();
    for (int d = t; d < OB_BINS; d += OBT) {
        uint32_t run = bin_base[d];
        for (int k = 0; k < OBT / 64; ++k) { const uint32_t c_ = wcnt[k * OB_BINS + d]; wcnt[k * OB_BINS + d] = (uint16_t)run; run += c_; }
    }
    __syncthreads();
    walk(true);
}

void launch_lsd_order(const LsdParams& p, int n_frames, const uint32_t* r_addr, const float* r_deg, const double* r_mod,
                      const double* r_cs, const double* r_sn, int* n_rec, const unsigned long long* maxgrad,
                      unsigned long long* sort_a, unsigned long long* sort_b, uint32_t* order_a, uint32_t* order_b,
                      int* norder, uint32_t* c_xy, float* c_deg, double* c_mod, double* c_cs, double* c_sn,
                      int* row_start, hipStream_t s)
{
    // the bit-plane form wherever the plane fits its LDS budget and the bins its counters (LF_ORDER_SORT=1: the sorting kernel of
    // rounds 1 - 3, for A/B; LF_ORDER_HBM=1: every problem down the bit-plane kernel's HBM branch, for the tests)
    static const bool old_form = getenv("LF_ORDER_SORT") != nullptr;
    static const int force_hbm = getenv("LF_ORDER_HBM") ? 1 : 0;
    const size_t Ps = (size_t)p.Hs * p.Ws;
    size_t blds = bitplane_lds_words(Ps) * 4;
    if (blds < (size_t)(OBT / 64) * OB_BINS * 2) blds = (size_t)(OBT / 64) * OB_BINS * 2;
    if (blds < (size_t)8 * OBT * 4) blds = (size_t)8 * OBT * 4;
    if (!old_form && p.n_bins <= OB_BINS && blds <= 40 * 1024 && Ps < ((size_t)1 << 20)) {
        hipLaunchKernelGGL(k_lsd_order_bm, dim3(n_frames * 3), dim3(OBT), blds, s, p, r_addr, r_deg, r_mod, r_cs, r_sn, n_rec, maxgrad,
                           sort_a, sort_b, order_a, order_b, norder, c_xy, c_deg, c_mod, c_cs, c_sn, row_start, force_hbm);
        return;
    }
    const size_t lds = ((size_t)2 * LDS_ITEMS + 2 * (size_t)(p.Hs + 2) + 2 * (OT / 64) * 32) * sizeof(uint32_t);
    // more than 64 KB of dynamic LDS needs the opt-in; the attribute is PER DEVICE, so it is set before every such launch
    // (a refusal surfaces through hipGetLastError() in the caller's LF_HIP_CHECK after the launch)
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_lsd_order), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_lsd_order, dim3(n_frames * 3), dim3(OT), lds, s, p, r_addr, r_deg, r_mod, r_cs, r_sn, n_rec, maxgrad,
                       sort_a, sort_b, order_a, order_b, norder, c_xy, c_deg, c_mod, c_cs, c_sn, row_start);
}

// ---------------------------------------------------------------------------------------------------------------
// K_lsd_label: connected components of a problem's defined pixels (8-adjacency), one workgroup per problem.
//
// LSD's region growing moves along 8-adjacent pixels with a defined gradient and looks at nothing but their USED
// flags (OpenCV lsd.cpp region_grow, restated in lsd_grow.h), so the components of that graph are independent
// sub-problems: growing them in any order, or at the same time, gives the sequential result as long as each keeps
// its own seeds in the global seed order.  k_lsd_grow hands the components of a problem to its waves, largest first.
// A component with fewer pixels than the smallest acceptable region can never yield a segment and is not listed.
//
// Union-find in LDS over the compact entries (lock-free: a root is only ever re-parented to a SMALLER index with
// atomicMin, path halving uses the same atomic, so every write moves a node closer to its final root): horizontal runs
// of consecutive pixels are linked to their first entry up front, then each entry is united with its up-left / up /
// up-right neighbours (found in the row lists).  The label of a
// component is its first entry in raster order.
//   c_label[e]      root of entry e                     (u16; problems of more than label_items entries: all 0)
//   comp_list[k]    roots of the components with >= min_reg_size pixels, by size descending, root ascending
//   comp_count      how many
// (the parent array lives in LDS and is addressed as such: a volatile access through a generic pointer is a flat load)
typedef __attribute__((address_space(3))) uint32_t uf_lds_u32;
typedef volatile uf_lds_u32 uf_lds_vu32;
// The union-find's storage.  UfLds: the LDS tables (problems of up to label_lds entries).  UfGlb (round 4): the same tables in the
// problem's region scratch in global memory, for the problems beyond that -- every access an agent-scope atomic (served by the L2:
// the vector L1 is not coherent with the atomics that re-parent nodes).  Slower per access, but such a workgroup asks for no
// dynamic LDS at all: with k_lsd_grow's workgroups of other batches filling every CU's LDS in 25 KB pieces, a labelling workgroup
// that needed 147 KB (24 k entries) only started once a whole CU had drained -- 18 ms in the queue against 0.6 ms of work
// (tools/pipe_overlap.py on camera frames), which capped the whole pipeline.
struct UfLds {                       // component sizes: u32 words
    uf_lds_vu32* P;
    __device__ __forceinline__ uint32_t get(uint32_t i) const { return P[i]; }
    __device__ __forceinline__ void set(uint32_t i, uint32_t v) const { P[i] = v; }
    __device__ __forceinline__ void add(uint32_t i, uint32_t v) const { (void)__hip_atomic_fetch_add((uf_lds_u32*)&P[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
};
typedef __attribute__((address_space(3))) uint16_t uf_lds_u16;
struct UfLds16 {                     // parents: u16 (an LDS problem has at most label_lds <= 65535 entries); min by compare-and-swap on the word
    volatile uf_lds_u16* P;
    __device__ __forceinline__ uint32_t get(uint32_t i) const { return P[i]; }
    __device__ __forceinline__ void set(uint32_t i, uint32_t v) const { P[i] = (uint16_t)v; }
    __device__ __forceinline__ uint32_t min(uint32_t i, uint32_t v) const
    {
        uf_lds_u32* w = (uf_lds_u32*)P + (i >> 1);
        const uint32_t sh = (i & 1u) * 16u;
        uint32_t cur = *(volatile uf_lds_u32*)w;
        for (;;) {
            const uint32_t old = (cur >> sh) & 0xffffu;
            if (v >= old) return old;
            uint32_t expect = cur;
            if (__hip_atomic_compare_exchange_strong(w, &expect, (cur & ~(0xffffu << sh)) | (v << sh), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return old;
            cur = expect;
        }
    }
};
struct UfGlb {
    uint32_t* P;
    __device__ __forceinline__ uint32_t get(uint32_t i) const { return __hip_atomic_load(P + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ void set(uint32_t i, uint32_t v) const { __hip_atomic_store(P + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ uint32_t min(uint32_t i, uint32_t v) const { return __hip_atomic_fetch_min(P + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ void add(uint32_t i, uint32_t v) const { (void)__hip_atomic_fetch_add(P + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
};
template <class UF>
__device__ __forceinline__ uint32_t uf_find(const UF& P, uint32_t x)
{
    uint32_t p = P.get(x);
    while (p != x) {
        const uint32_t g = P.get(p);
        if (g != p) (void)P.min(x, g);                                 // path halving
        x = p;
        p = g;
    }
    return x;
}

template <class UF>
__device__ __forceinline__ void uf_unite(const UF& P, uint32_t a, uint32_t b)
{
    for (;;) {
        a = uf_find(P, a);
        b = uf_find(P, b);
        if (a == b) return;
        if (a < b) { const uint32_t t = a; a = b; b = t; }            // a > b: hang a below b
        const uint32_t old = P.min(a, b);
        if (old == a) return;                                          // a was still a root: done
        a = old;                                                       // somebody re-parented a meanwhile: go on from there
    }
}

constexpr int LT = LF_LABEL_THREADS;
#ifndef LF_COMP_TOP
#define LF_COMP_TOP 8
#endif
constexpr int kCompTop = LF_COMP_TOP;       // components that get a pass over the seed list of their own; the rest share one

// GLB = false: the problems of up to label_lds entries, tables in LDS (4 bytes per entry: u16 parent + u16 x, later u16 size).
// GLB = true: the larger ones (up to label_items), tables in the problem's region scratch, neighbours through the bit plane.
template <bool GLB>
__device__ __forceinline__ void lsd_label_problem(const LsdParams& p, const int* __restrict__ norder, const uint32_t* __restrict__ c_xy,
                                                  const int* __restrict__ row_start, uint16_t* __restrict__ c_label,
                                                  uint16_t* __restrict__ comp_list, int* __restrict__ comp_count, int* __restrict__ comp_key, int comp_cap,
                                                  uint32_t* __restrict__ scratch, size_t scratch_stride)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn_lds[];
    const int LS = p.label_lds;
    __shared__ uint16_t roots[kCompCap];
    __shared__ int n_roots, rest_total;
    const int pc = blockIdx.x, t = threadIdx.x;
    const size_t Ps = (size_t)p.Hs * p.Ws, o = (size_t)pc * p.rec_cap;
    const int n = norder[pc];
    uint16_t* lab = c_label + o;
    uint16_t* list = comp_list + (size_t)pc * kCompCap;
    if (n == 0) { if (t == 0) { comp_count[pc] = 0; comp_key[pc] = 0; } return; }
    if (n > p.label_items) {
        // beyond the 16-bit labels / the scratch: one component = the whole problem, a valid labelling (k_lsd_grow then grows it
        // with one wave)
        for (int e = t; e < n; e += LT) lab[e] = 0;
        if (t == 0) { list[0] = 0; comp_count[pc] = 1; comp_key[pc] = n; }
        return;
    }
    // parent [n]; the component sizes, two u16 counters per word [(n + 1) / 2] (LDS: where the x lists were)
    const uint16_t* xs_lds = reinterpret_cast<const uint16_t*>(dyn_lds + LS / 2);
    uint32_t* gtab = scratch + (size_t)pc * scratch_stride;
    typename std::conditional<GLB, UfGlb, UfLds16>::type parent;
    typename std::conditional<GLB, UfGlb, UfLds>::type csize2;
    if constexpr (GLB) { parent.P = gtab; csize2.P = gtab + ((n + 1) & ~1); }
    else { parent.P = (volatile uf_lds_u16*)dyn_lds; csize2.P = (uf_lds_vu32*)(dyn_lds + LS / 2); }
    auto xs = [&](int k) -> int { if constexpr (GLB) return (int)(c_xy[o + k] & 0xffffu); else return (int)xs_lds[k]; };
    const int* rs = row_start + (size_t)pc * (p.Hs + 1);
    if (t == 0) { n_roots = 0; rest_total = 0; }
    if constexpr (GLB) {
        // The large problems find their neighbours in the bit plane (lsd_bitplane.h: 9 bytes of LDS per 64 pixels whatever the
        // problem's size), and work run by run: a pixel whose left neighbour is defined belongs to that pixel's run, and the
        // run already touches everything above that the pixel's up-left and up neighbours belong to -- so only the first pixel
        // of a run looks at all three pixels above, the others only at a run that BEGINS up-right of them.
        __shared__ int wave_tot[LT / 64];
        bitplane_build<LT>(dyn_lds, c_xy + o, n, p.Ws, Ps, wave_tot);
        __syncthreads();
        const unsigned long long* bits64 = reinterpret_cast<const unsigned long long*>(dyn_lds);
        const uint16_t* pref = reinterpret_cast<const uint16_t*>(dyn_lds + 2 * bitplane_words(Ps));
        auto bit = [&](int pos) -> bool { return (bits64[pos >> 6] >> (pos & 63)) & 1ull; };
        auto rank = [&](int pos) -> uint32_t {
            const int w = pos >> 6;
            const unsigned long long lo = bits64[w & ~1], cur = bits64[w];
            return (uint32_t)pref[w >> 1] + ((w & 1) ? (uint32_t)__builtin_popcountll(lo) : 0u) + (uint32_t)__builtin_popcountll(cur & ((1ull << (pos & 63)) - 1ull));
        };
        for (int e = t; e < n; e += LT) {
            const uint32_t xy = c_xy[o + e];
            const int x = (int)(xy & 0xffffu), y = (int)(xy >> 16), pos = y * p.Ws + x;
            // the defined pixels directly to the left, inside the row: the run's first entry is that many entries back
            int back = 0;
            while (back < x && bit(pos - back - 1)) ++back;
            parent.set((uint32_t)e, (uint32_t)(e - back));
        }
        __syncthreads();
        for (int e = t; e < n; e += LT) {
            const uint32_t xy = c_xy[o + e];
            const int x = (int)(xy & 0xffffu), y = (int)(xy >> 16);
            if (y == 0) continue;
            const int up = (y - 1) * p.Ws + x;
            const bool head = x == 0 || !bit(up + p.Ws - 1);
            const bool uL = x > 0 && bit(up - 1), uC = bit(up), uR = x + 1 < p.Ws && bit(up + 1);
            if (head) {
                if (uL) uf_unite(parent, (uint32_t)e, rank(up - 1));
                else if (uC) uf_unite(parent, (uint32_t)e, rank(up));
                if (uR && !uC) uf_unite(parent, (uint32_t)e, rank(up + 1));
            } else if (uR && !uC) uf_unite(parent, (uint32_t)e, rank(up + 1));
        }
        __syncthreads();
    } else {
        for (int e = t; e < n; e += LT) reinterpret_cast<uint16_t*>(dyn_lds + LS / 2)[e] = (uint16_t)(c_xy[o + e] & 0xffffu);
        __syncthreads();
    // horizontal runs first, without atomics: every entry starts as a child of the first entry of its run of
    // consecutive x (a forest of depth one whose roots are the smallest indices), so the union phase below only has
    // to stitch runs of adjacent rows together
    for (int e = t; e < n; e += LT) {
        const int y = (int)(c_xy[o + e] >> 16);
        const int first = rs[y];
        int h = e;
        int xh = xs(h);
        while (h > first) { const int xp = xs(h - 1); if (xp + 1 != xh) break; --h; xh = xp; }
        parent.set((uint32_t)e, (uint32_t)h);
    }
    __syncthreads();
    for (int e = t; e < n; e += LT) {
        const uint32_t xy = c_xy[o + e];
        const int x = (int)(xy & 0xffffu), y = (int)(xy >> 16);
        if (y > 0) {
            int lo = rs[y - 1];
            const int end = rs[y];
            int hi = end;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (xs(mid) < x - 1) lo = mid + 1; else hi = mid; }
            for (int k = lo; k < end && k < lo + 3 && xs(k) <= x + 1; ++k) uf_unite(parent, (uint32_t)e, (uint32_t)k);
        }
    }
    __syncthreads();
    }
    for (int i = t; i < (n + 1) / 2; i += LT) csize2.set((uint32_t)i, 0u);     // (LDS: the x lists are no longer needed)
    __syncthreads();
    for (int e = t; e < n; e += LT) {
        const uint32_t r = uf_find(parent, (uint32_t)e);
        lab[e] = (uint16_t)r;
        csize2.add(r >> 1, 1u << (16 * (r & 1u)));              // counts stay below 2^16 (n < 65536): no carry between the halves
    }
    __syncthreads();
    auto csize = [&](uint32_t r) -> uint32_t { return (csize2.get(r >> 1) >> (16 * (r & 1u))) & 0xffffu; };
    const uint32_t minsz = p.min_reg_size > 1 ? (uint32_t)p.min_reg_size : 1u;
    for (int e = t; e < n; e += LT)
        if (csize((uint32_t)e) >= minsz) {                       // only roots have a count
            const int k = atomicAdd(&n_roots, 1);
            if (k < comp_cap) roots[k] = (uint16_t)e;
        }
    __syncthreads();
    const int C = n_roots;
    if (C > comp_cap) {
        // more eligible components than the list holds (tiny min_reg_size): fall back to one component
        for (int e = t; e < n; e += LT) lab[e] = 0;
        if (t == 0) { list[0] = 0; comp_count[pc] = 1; comp_key[pc] = n; }
        return;
    }
    for (int i = t; i < C; i += LT) {
        const uint32_t ri = roots[i], si = csize(ri);
        int rank = 0;
        for (int j = 0; j < C; ++j) {
            const uint32_t rj = roots[j], sj = csize(rj);
            rank += (sj > si || (sj == si && rj < ri)) ? 1 : 0;
        }
        list[rank] = (uint16_t)ri;
        parent.set(ri, (uint32_t)rank);                          // (the union-find is done with: a root's slot now holds its rank)
        if (rank == 0) comp_key[pc] = (int)si;                   // launch-order key (k_lsd_rank): the largest component = the longest wave
        if (rank >= kCompTop) atomicAdd(&rest_total, (int)si);
    }
    __syncthreads();
    if (C <= kCompTop) {
        if (t == 0) { comp_count[pc] = C; if (C == 0) comp_key[pc] = 0; }
        return;
    }
    // Many components (speckle, texture): every listed component costs the wave that takes it a pass over the whole seed list,
    // so only the kCompTop largest stand alone; the others are grown TOGETHER, as one sub-problem under the label of the
    // largest of them (independent components may be worked through in any interleaving -- one wave going through several of
    // them in seed order is the sequential algorithm restricted to them).  The group comes first when it is the biggest piece.
    __threadfence_block();
    const uint16_t rest_root = list[kCompTop];
    for (int e = t; e < n; e += LT) {
        const uint32_t r = lab[e];
        if (csize(r) >= minsz && parent.get(r) >= (uint32_t)kCompTop) lab[e] = rest_root;
    }
    if (t == 0) {
        const int rest = rest_total;
        const int top0 = (int)csize((uint32_t)list[0]);
        if (rest > top0) {
            for (int k = kCompTop; k > 0; --k) list[k] = list[k - 1];
            list[0] = rest_root;
            comp_key[pc] = rest;
        }
        comp_count[pc] = kCompTop + 1;
    }
}

// Launch order of k_lsd_grow's workgroups: problems with the largest connected component first (k_lsd_label's comp_key;
// longest first: when the chip is
// full of another batch's growing waves, the workgroups of this launch start as slots free up, and the long problems
// should not be the last to start).  rank by counting: n_prob <= 768.
__global__ __launch_bounds__(256) void k_lsd_rank(int n_prob, const int* __restrict__ norder, int* __restrict__ perm)
{
    // one workgroup of four waves (1024 threads waited for a CU with sixteen free wave slots: see k_hysteresis_cols)
    __shared__ int nd[1024];
    if (n_prob > 1024) {                                     // (never: 3 x 256 frames; kept general)
        for (int i = threadIdx.x; i < n_prob; i += 256) perm[i] = i;
        return;
    }
    for (int i = threadIdx.x; i < n_prob; i += 256) nd[i] = norder[i];
    __syncthreads();
    for (int i = threadIdx.x; i < n_prob; i += 256) {
        const int mine = nd[i];
        int rank = 0;
        for (int j = 0; j < n_prob; ++j) rank += (nd[j] > mine || (nd[j] == mine && j < i)) ? 1 : 0;
        perm[rank] = i;
    }
}

void launch_lsd_rank(int n_prob, const int* norder, int* perm, hipStream_t s)
{
    hipLaunchKernelGGL(k_lsd_rank, dim3(1), dim3(256), 0, s, n_prob, norder, perm);
}

// one launch, the form chosen per problem (both fit 24 KB of dynamic LDS; two launches cost a busy pipeline ~2 ms of a batch's latency)
__global__ __launch_bounds__(LT) void k_lsd_label(LsdParams p, const int* __restrict__ norder, const uint32_t* __restrict__ c_xy,
                                                  const int* __restrict__ row_start, uint16_t* __restrict__ c_label,
                                                  uint16_t* __restrict__ comp_list, int* __restrict__ comp_count, int* __restrict__ comp_key, int comp_cap,
                                                  uint32_t* __restrict__ scratch, size_t scratch_stride, int plane_fits)
{
    const int n = norder[blockIdx.x];
    if (n <= p.label_lds)
        lsd_label_problem<false>(p, norder, c_xy, row_start, c_label, comp_list, comp_count, comp_key, comp_cap, scratch, scratch_stride);
    else if (plane_fits)
        lsd_label_problem<true>(p, norder, c_xy, row_start, c_label, comp_list, comp_count, comp_key, comp_cap, scratch, scratch_stride);
    else {
        // (images whose bit plane exceeds the launch's LDS: 1080p) one component = the whole problem
        const size_t o = (size_t)blockIdx.x * p.rec_cap;
        for (int e = threadIdx.x; e < n; e += LT) c_label[o + e] = 0;
        if (threadIdx.x == 0) { comp_list[(size_t)blockIdx.x * kCompCap] = 0; comp_count[blockIdx.x] = 1; comp_key[blockIdx.x] = n; }
    }
}

void launch_lsd_label(const LsdParams& p, int n_frames, const int* norder, const uint32_t* c_xy, const int* row_start,
                      uint16_t* c_label, uint16_t* comp_list, int* comp_count, int* comp_key, uint32_t* scratch, hipStream_t s)
{
    size_t lds = (size_t)p.label_lds * (2 + 2);
    const size_t plane = bitplane_lds_words((size_t)p.Hs * p.Ws) * 4;
    const int plane_fits = plane <= 150 * 1024 ? 1 : 0;       // (1080p: 124 KB, one workgroup per CU -- as the 144 KB tables of round 3 were)
    if (plane_fits && plane > lds) lds = plane;
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_lsd_label), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // LF_DIAG_COMP_CAP: a smaller component list, so that tests reach the "more components than the list holds" fallback
    static const int comp_cap = getenv("LF_DIAG_COMP_CAP") ? max(1, min(kCompCap, atoi(getenv("LF_DIAG_COMP_CAP")))) : kCompCap;
    const size_t stride = lsd_grow_reg_stride(p);               // the region scratch is free until k_lsd_grow runs
    hipLaunchKernelGGL(k_lsd_label, dim3(n_frames * 3), dim3(LT), lds, s, p, norder, c_xy, row_start, c_label, comp_list, comp_count, comp_key, comp_cap, scratch, stride, plane_fits);
}

// Debug only: dense angle / magnitude planes rebuilt from the compact arrays (NOTDEF / 0 elsewhere).
__global__ void k_lsd_dense_debug(LsdParams p, const int* __restrict__ norder, const uint32_t* __restrict__ c_xy,
                                  const float* __restrict__ c_deg, const double* __restrict__ c_mod,
                                  float* __restrict__ ang, double* __restrict__ mod)
{
    const int pc = blockIdx.x;
    const size_t Ps = (size_t)p.Hs * p.Ws, o = (size_t)pc * Ps, oc = (size_t)pc * p.rec_cap;     // (dense planes | compact arrays)
    for (size_t i = threadIdx.x; i < Ps; i += blockDim.x) { ang[o + i] = kNotDef; mod[o + i] = 0.0; }
    __syncthreads();
    const int n = norder[pc];
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const uint32_t xy = c_xy[oc + e];
        const size_t a = (size_t)(xy >> 16) * p.Ws + (xy & 0xffffu);
        ang[o + a] = c_deg[oc + e];
        mod[o + a] = c_mod[oc + e];
    }
}

void launch_lsd_dense_debug(const LsdParams& p, int n_frames, const int* norder, const uint32_t* c_xy, const float* c_deg,
                            const double* c_mod, float* ang, double* mod, hipStream_t s)
{
    hipLaunchKernelGGL(k_lsd_dense_debug, dim3(n_frames * 3), dim3(1024), 0, s, p, norder, c_xy, c_deg, c_mod, ang, mod);
}

}  // namespace lf
