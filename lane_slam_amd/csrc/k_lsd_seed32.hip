// K_lsd_seed32: the LSD seed order of OpenCV 3.2 ... 3.4.5 (the reference's stack: ROS Kinetic = 3.3.1) on the device.
//
// Reference call site: /root/reference/src/line_detector/include/line_detector/line_detector_lsd.py:64-72 (cv2
// createLineSegmentDetector(REFINE_ADV).detect).  From 3.2 on ll_angle pushes one normPoint {x, y, norm = int(modgrad *
// bin_coef)} per pixel of the (H-1) x (W-1) gradient image in raster order -- pixels WITHOUT a defined gradient included -- and
// orders the list with   std::sort(ordered_points.begin(), ordered_points.end(), compare_norm)   (n1.norm > n2.norm).
// std::sort is not stable: inside a bin the seeds come in whatever order libstdc++'s introsort leaves them in, and region
// growing depends on the seed order.  (OpenCV 3.0 / 3.1 kept per-bin lists in raster order: k_lsd_order.hip, the other setting.)
//
// What libstdc++ does (bits/stl_algo.h: __sort = __introsort_loop(first, last, 2 * floor(log2 n)) + __final_insertion_sort):
//   introsort loop   while a range holds more than 16 elements: (depth limit used up: heap sort the range, done); the median of
//                    first + 1, middle, last - 1 goes to `first` (__move_median_to_first), __unguarded_partition of
//                    [first + 1, last) around it returns `cut`; recurse into [cut, last), go on with [first, cut).
//   final insertion  a plain insertion sort of the whole array = a STABLE sort by key of what the loop left.
// The ranges of the loop are disjoint, so they may be partitioned in any order or at the same time; only each partition has
// to move exactly the elements the sequential one moves.  __unguarded_partition:
//       for (;;) { while (comp(*first, pivot)) ++first;  --last;  while (comp(pivot, *last)) --last;
//                  if (!(first < last)) return first;  iter_swap(first, last);  ++first; }
// The left scan only ever stops on elements with !comp(x, pivot) (key <= pivot's: "L" elements), the right scan on elements
// with !comp(pivot, x) (key >= pivot's: "R"), and until the pointers cross neither sees an element the other has moved.  So
// the k-th L element from the left is swapped with the k-th R element from the right for k = 1 .. K, K = the number of k
// with L_k < R_k, and cut = L_1 when K = 0, else min(L_{K+1}, R_K)  (after the K-th swap the left pointer stops at the next
// L element or at R_K, which now holds one).  Ranks come from ballots and prefix sums; the partner R_k of a swapped L_k is
// found by a binary search in the suffix counts and a bit select in its row's ballot word.
//
// One workgroup per problem (frame, colour).  Elements are u32: bin << 20 | payload (compact index + 1 of a pixel with a
// defined gradient, 0 for the others: only seeds need to be told apart, the sort's decisions depend on keys alone).
// A RANGE WITHOUT A SEED IS NEVER PARTITIONED: the loop only permutes a range within itself, so what it does to a range that
// holds no seed cannot be seen in the result.  That prunes most of the work: the gradient image is flat almost everywhere (bin
// 0), the median of three is then 0, a partition around 0 leaves nothing but zeros to the right of the cut -- the top levels
// shed half of their elements each, and the ranges that remain are the few thousand pixels near an edge.
//   phase 0  the array: zeros (flat pixels: bin 0), the defined pixels from the compact arrays of k_lsd_order, the
//            undefined pixels with a non-zero gradient from k_lsd_grad's "low" records
//   phase 1  ranges of more than kSmall elements, in global memory: the whole workgroup partitions one range at a time in
//            four streaming passes -- (1) L / R ballots and counts of every 64-element row, eight rows in flight per wave;
//            (2) prefix / suffix sums of the row counts; (3) every swapped L and R element publishes its place under its rank;
//            (4) the pairs are swapped, eight in flight per lane.  No pass waits for a search or for another lane's element.
//   phase 2  every wave takes ranges of at most kSmall elements from a list, copies one into LDS and works off its whole
//            subtree there, alone (same four passes, private stack), then copies it back
//   phase 3  the seeds in the order the loop left them, compacted, then the final insertion sort as stable 4-bit counting
//            passes over the bin key -> order_a, the seed list k_lsd_grow reads
// The heap sort of a range that used up the depth limit (never seen on image data; tested with adversarial keys through
// lf_debug_std_sort) is libstdc++'s __heap_select + __sort_heap replayed by one lane.
#include <cstdlib>
#include "common.h"

namespace lf {

// Diagnostic build only (-DLF_SEED_STAMPS): per-phase cycle counts of the first problems, printed from the kernel
#ifdef LF_SEED_STAMPS
#define SEED_T(v) const long long v = (long long)wall_clock64()
#else
#define SEED_T(v) do { } while (0)
#endif

// Workgroup shape (round 4, end): 256 threads, four waves in phase 2, LDS blocks of 4096 elements.  The first form ran 1024 threads
// with 64 KB of LDS per problem: sixteen waves on one CU and that much LDS only come free together when a CU drains, and the
// kernel queued behind other batches' region growing -- smaller is faster even ALONE (the
// partitions are chains of dependent round trips, not throughput), same-call sweep, frames/s with lsd.seed_order = opencv32 at
// 640x480 / 160x120:   1024 thr, 8 waves, 8192: 65.8 k / 198 k      512, 8, 8192: 77.4 k / 238 k      256, 4, 4096: 74 k / 257 k
//                       256, 4, 2048: 72 k / 259 k      128, 2, 2048: 56 k / 271 k.   Every shape passes tests/test_gpu_seed_order.py.
#ifndef LF_SEED_THREADS
#define LF_SEED_THREADS 256
#endif
#ifndef LF_SEED_WAVES2
#define LF_SEED_WAVES2 4
#endif
#ifndef LF_SEED_BLOCK
#define LF_SEED_BLOCK 4096
#endif
constexpr int ST = LF_SEED_THREADS;  // threads
constexpr int SW = ST / 64;
constexpr int SW2 = LF_SEED_WAVES2;  // waves that work in phase 2 (each with a private LDS range)
constexpr int kSmall = 1024;         // ranges up to this size are one wave's work, in LDS (16 rows)
constexpr int kBlock = LF_SEED_BLOCK; // ranges up to this size are copied into LDS and partitioned there by the whole workgroup
constexpr int kSortThreshold = 16;   // libstdc++ _S_threshold
constexpr int kMaxLdsBytes = 150 * 1024;
constexpr int SNB = 16;              // buckets of the final counting passes
constexpr int kU = 8;                // rows / pairs in flight per wave / lane in the streaming passes
// one wave's private LDS in phase 2, 32-bit words: the range, the two place lists (u16), row tables, accumulators, range stack
constexpr int kWaveWords = kSmall + kSmall / 2 + 2 * 18 * 2 + 18 + 20 + 8 + 128;
// phase 1b (a block of <= kBlock elements in LDS): [block][aliased: the workgroup's place lists (u16) | 8 waves' private lists, tables,
// stacks][the workgroup's row tables][the block's list of small ranges]
constexpr int kBlkWaveWords = kSmall / 2 + 2 * 18 * 2 + 18 + 20 + 8 + 128;
constexpr int kBlkX = SW2 * kBlkWaveWords > kBlock / 2 ? SW2 * kBlkWaveWords : kBlock / 2;
constexpr int kBlkRows = kBlock / 64 + 2;
constexpr int kBlkWords = kBlock + kBlkX + (kBlkRows * 6 + 8) + 2 * (kBlock / 16);

__device__ __forceinline__ uint32_t key_of(uint32_t v) { return v >> 20; }
// compare_norm(a, b) = a.norm > b.norm
__device__ __forceinline__ bool comp(uint32_t a, uint32_t b) { return key_of(a) > key_of(b); }

// Everything the partition passes keep in LDS is addressed AS LDS (address space 3): through a generic pointer every table access
// is a flat instruction -- the long way to LDS, and a wait for whatever global access is in flight (v2 of this kernel spent most
// of its time there: 50 us per partition).
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
typedef __attribute__((address_space(3))) int lds_i32;
template <typename T> __device__ __forceinline__ T* as_lds(void* generic) { return (T*)(__attribute__((address_space(3))) void*)generic; }
__device__ __forceinline__ void lds_add(lds_i32* p, int v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_min(lds_i32* p, int v) { (void)__hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_or(lds_i32* p, int v) { (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

__device__ __forceinline__ int wave_incl_scan_i(int v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int n = __shfl_up(v, d);
        if (lane >= d) v += n;
    }
    return v;
}

// position of the t-th (1-based) set bit of b counted from bit 63 downwards: the largest p with popcount(b >> p) >= t
__device__ __forceinline__ int select_from_top(unsigned long long b, int t)
{
    int pos = 0;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1)
        if (__popcll(b >> (pos + s)) >= t) pos += s;
    return pos;
}

// __move_median_to_first(result = f, a = f + 1, b = mid, c = l - 1)
__device__ __forceinline__ void median_to_first(uint32_t* E, int f, int l)
{
    const int ia = f + 1, ib = f + (l - f) / 2, ic = l - 1;
    const uint32_t a = E[ia], b = E[ib], c = E[ic];
    int pick;
    if (comp(a, b)) pick = comp(b, c) ? ib : (comp(a, c) ? ic : ia);
    else pick = comp(a, c) ? ia : (comp(b, c) ? ic : ib);
    const uint32_t r = E[f], p = E[pick];
    E[f] = p;
    E[pick] = r;
}

// ---- libstdc++'s heap sort of [f, l) (std::__partial_sort(first, last, last)), one lane -----------------------------------
__device__ void push_heap_(uint32_t* A, int hole, int top, uint32_t value)
{
    int parent = (hole - 1) / 2;
    while (hole > top && comp(A[parent], value)) {
        A[hole] = A[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    A[hole] = value;
}
__device__ void adjust_heap_(uint32_t* A, int hole, int len, uint32_t value)
{
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (comp(A[child], A[child - 1])) --child;
        A[hole] = A[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        A[hole] = A[child - 1];
        hole = child - 1;
    }
    push_heap_(A, hole, top, value);
}
__device__ __noinline__ void heap_sort_range(uint32_t* E, int f, int l)
{
    uint32_t* A = E + f;
    const int len = l - f;
    if (len < 2) return;
    for (int parent = (len - 2) / 2;; --parent) {            // __make_heap
        adjust_heap_(A, parent, len, A[parent]);
        if (parent == 0) break;
    }
    for (int last = len - 1; last > 0; --last) {              // __sort_heap: __pop_heap(first, last, last)
        const uint32_t value = A[last];
        A[last] = A[0];
        adjust_heap_(A, 0, last, value);
    }
}

// Row tables of one partition: BL / BR the ballots of the L and R elements of every 64-element row of [lo, hi), PL[r] the L
// elements in rows < r, SX[r] the R elements in rows >= r (SX[R] = 0).
// acc: [0] K  [1] first L  [2] first L that stays  [3] seeds in the range
template <bool COOP>
__device__ __forceinline__ void team_sync()
{
    if (COOP) __syncthreads();
    else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
}

// Passes 2 - 4 and the cut, given the row ballots of [lo, hi) in BL / BR (pass 1 differs between the two users).  EP: the
// elements (global or LDS), LP: the place lists (u32 in global scratch / u16 in LDS), positions relative to `org`.
template <bool COOP, typename EP, typename LP>
__device__ __forceinline__ int partition_tail(EP E, int org, int lo, int hi, lds_u64* BL, lds_u64* BR, lds_i32* PL, lds_i32* SX,
                                              lds_i32* acc, LP Lpos, LP Rpos, int w, int nw, int lane, int tid, int nthreads)
{
    const int R = (hi - lo + 63) >> 6;
    const unsigned long long le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);      // lanes <= this one
    const unsigned long long lt = (1ull << lane) - 1ull;                                // lanes < this one
    // (2) exclusive prefix of the L counts (first wave of the team), inclusive suffix of the R counts (its second wave)
    if (w == 0) {
        int carry = 0;
        for (int r0 = 0; r0 < R; r0 += 64) {
            const int r = r0 + lane;
            const int c = r < R ? PL[r] : 0;
            const int inc = wave_incl_scan_i(c, lane);
            if (r < R) PL[r] = carry + inc - c;
            carry += __shfl(inc, 63);
        }
    }
    if (w == (COOP ? 1 : 0)) {
        int carry = 0;
        for (int r0 = 0; r0 < R; r0 += 64) {
            const int r = R - 1 - (r0 + lane);                 // from the last row backwards
            const int c = r >= 0 ? SX[r] : 0;
            const int inc = wave_incl_scan_i(c, lane);
            if (r >= 0) SX[r] = carry + inc;
            carry += __shfl(inc, 63);
        }
        if (lane == 0) SX[R] = 0;
    }
    team_sync<COOP>();
    // (3) ranks; an L element of rank k is swapped iff at least k R elements lie to its right (L_k < R_k), an R element of rank
    // k iff at least k L elements lie to its left; both publish their place under their rank
    int wK = 0, wFirstL = 0x7fffffff, wFirstStay = 0x7fffffff;
    for (int r = w; r < R; r += nw) {
        const int i = lo + r * 64 + lane;
        const unsigned long long bl = BL[r], br = BR[r];
        const bool isL = (bl >> lane) & 1ull, isR = (br >> lane) & 1ull;
        const int kl = PL[r] + __popcll(bl & le), r_right = SX[r + 1] + __popcll(br & ~le);
        const int kr = SX[r + 1] + __popcll(br >> lane), l_left = PL[r] + __popcll(bl & lt);
        const bool swl = isL && r_right >= kl, swr = isR && l_left >= kr;
        if (swl) Lpos[kl - 1] = i - org;
        if (swr) Rpos[kr - 1] = i - org;
        const unsigned long long bs = __ballot(swl);
        const unsigned long long un = bl & ~bs;                                 // L elements of the row that stay
        wK += __popcll(bs);
        if (bl) wFirstL = min(wFirstL, lo + r * 64 + __ffsll((long long)bl) - 1);
        if (un) wFirstStay = min(wFirstStay, lo + r * 64 + __ffsll((long long)un) - 1);
    }
    if (lane == 0) {
        if (wK) lds_add(&acc[0], wK);
        if (wFirstL != 0x7fffffff) lds_min(&acc[1], wFirstL);
        if (wFirstStay != 0x7fffffff) lds_min(&acc[2], wFirstStay);
    }
    team_sync<COOP>();
    // (4) the swaps
    const int K = acc[0];
    for (int k0 = tid; k0 < K; k0 += nthreads * kU) {
        int pi[kU], pj[kU];
        uint32_t a[kU], b[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int k = k0 + u * nthreads;
            pi[u] = k < K ? org + (int)Lpos[k] : -1;
            pj[u] = k < K ? org + (int)Rpos[k] : -1;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) if (pi[u] >= 0) { a[u] = E[pi[u]]; b[u] = E[pj[u]]; }
#pragma unroll
        for (int u = 0; u < kU; ++u) if (pi[u] >= 0) { E[pi[u]] = b[u]; E[pj[u]] = a[u]; }
    }
    const int cut = K == 0 ? acc[1] : min(acc[2], org + (int)Rpos[K - 1]);
    team_sync<COOP>();                                                           // the tables are free again
    return cut;
}

// __unguarded_partition of [f + 1, l) around E[f] (already the median) in global memory by the whole workgroup.  Returns the
// cut, or -1 when the range [f, l) holds no seed (nothing was moved then, and nothing needs to be).
__device__ __forceinline__ int partition_global(uint32_t* E, int f, int l, lds_u64* BL, lds_u64* BR, lds_i32* PL, lds_i32* SX,
                                                lds_i32* acc, uint32_t* Lpos, uint32_t* Rpos, int w, int lane)
{
    const int lo = f + 1, hi = l;
    const int R = (hi - lo + 63) >> 6;
    const uint32_t pivot = E[f];
    if (threadIdx.x == 0) { acc[0] = 0; acc[1] = 0x7fffffff; acc[2] = 0x7fffffff; acc[3] = (pivot & 0xfffffu) != 0u; }
    __syncthreads();
    // (1) eight rows in flight per wave
    for (int r0 = w * kU; r0 < R; r0 += SW * kU) {
        uint32_t v[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) { const int i = lo + (r0 + u) * 64 + lane; v[u] = i < hi ? E[i] : 0xffffffffu; }
        int seeds = 0;
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int r = r0 + u;
            if (r < R) {
                const bool valid = lo + r * 64 + lane < hi;
                const unsigned long long bl = __ballot(valid && !comp(v[u], pivot));
                const unsigned long long br = __ballot(valid && !comp(pivot, v[u]));
                seeds += __popcll(__ballot(valid && (v[u] & 0xfffffu) != 0u));
                if (lane == 0) { BL[r] = bl; BR[r] = br; PL[r] = __popcll(bl); SX[r] = __popcll(br); }
            }
        }
        if (lane == 0 && seeds) lds_add(&acc[3], seeds);
    }
    __syncthreads();
    if (acc[3] == 0) { __syncthreads(); return -1; }
    return partition_tail<true>(E, 0, lo, hi, BL, BR, PL, SX, acc, Lpos, Rpos, w, SW, lane, (int)threadIdx.x, ST);
}

// the same for a range [f, l) of a wave's LDS copy D (positions relative to the copy), one wave alone
__device__ __forceinline__ int partition_lds(lds_u32* D, int f, int l, lds_u64* BL, lds_u64* BR, lds_i32* PL, lds_i32* SX,
                                             lds_i32* acc, lds_u16* Lpos, lds_u16* Rpos, int lane)
{
    const int lo = f + 1, hi = l;
    const int R = (hi - lo + 63) >> 6;
    const uint32_t pivot = D[f];
    if (lane == 0) { acc[0] = 0; acc[1] = 0x7fffffff; acc[2] = 0x7fffffff; }
    int seeds = (pivot & 0xfffffu) != 0u;
    for (int r = 0; r < R; ++r) {
        const int i = lo + r * 64 + lane;
        const bool valid = i < hi;
        const uint32_t v = valid ? D[i] : 0u;
        const unsigned long long bl = __ballot(valid && !comp(v, pivot));
        const unsigned long long br = __ballot(valid && !comp(pivot, v));
        seeds += __popcll(__ballot(valid && (v & 0xfffffu) != 0u));
        if (lane == 0) { BL[r] = bl; BR[r] = br; PL[r] = __popcll(bl); SX[r] = __popcll(br); }
    }
    team_sync<false>();
    if (seeds == 0) return -1;
    return partition_tail<false>(D, 0, lo, hi, BL, BR, PL, SX, acc, Lpos, Rpos, 0, 1, lane, lane, 64);
}

__device__ __forceinline__ void median_to_first_lds(lds_u32* E, int f, int l)
{
    const int ia = f + 1, ib = f + (l - f) / 2, ic = l - 1;
    const uint32_t a = E[ia], b = E[ib], c = E[ic];
    int pick;
    if (comp(a, b)) pick = comp(b, c) ? ib : (comp(a, c) ? ic : ia);
    else pick = comp(a, c) ? ia : (comp(b, c) ? ic : ib);
    const uint32_t r = E[f], p = E[pick];
    E[f] = p;
    E[pick] = r;
}

// The introsort loop over E[0, n): phases 1 and 2 of the header.  scratch: global, 3 n / 4 + 64 u64 entries (the list of small
// ranges, then the two place lists of the global partitions).
// __unguarded_partition of [f + 1, l) of an LDS block by the whole workgroup (pass 1 on LDS, then the shared tail)
__device__ __forceinline__ int partition_lds_coop(lds_u32* D, int f, int l, lds_u64* BL, lds_u64* BR, lds_i32* PL, lds_i32* SX,
                                                  lds_i32* acc, lds_u16* Lpos, lds_u16* Rpos, int w, int lane)
{
    const int lo = f + 1, hi = l;
    const int R = (hi - lo + 63) >> 6;
    const uint32_t pivot = D[f];
    if (threadIdx.x == 0) { acc[0] = 0; acc[1] = 0x7fffffff; acc[2] = 0x7fffffff; acc[3] = (pivot & 0xfffffu) != 0u; }
    __syncthreads();
    int seeds = 0;
    for (int r = w; r < R; r += SW) {
        const int i = lo + r * 64 + lane;
        const bool valid = i < hi;
        const uint32_t v = valid ? D[i] : 0u;
        const unsigned long long bl = __ballot(valid && !comp(v, pivot));
        const unsigned long long br = __ballot(valid && !comp(pivot, v));
        seeds += __popcll(__ballot(valid && (v & 0xfffffu) != 0u));
        if (lane == 0) { BL[r] = bl; BR[r] = br; PL[r] = __popcll(bl); SX[r] = __popcll(br); }
    }
    if (lane == 0 && seeds) lds_add(&acc[3], seeds);
    __syncthreads();
    if (acc[3] == 0) { __syncthreads(); return -1; }
    return partition_tail<true>(D, 0, lo, hi, BL, BR, PL, SX, acc, Lpos, Rpos, w, SW, lane, (int)threadIdx.x, ST);
}

// one wave works off the whole subtree of the range [f, l) of an LDS array alone (private tables, lists and stack)
__device__ __forceinline__ void wave_subtree(lds_u32* D, uint32_t* D_generic, int f, int l, int depth, lds_u64* wBL, lds_u64* wBR,
                                             lds_i32* wPL, lds_i32* wSX, lds_i32* wacc, lds_u16* Lp, lds_u16* Rp, lds_i32* stack, int lane)
{
    int sp = 0;
    for (;;) {
        while (l - f > kSortThreshold) {
            if (depth == 0) {
                if (lane == 0) heap_sort_range(D_generic, f, l);
                team_sync<false>();
                break;
            }
            --depth;
            if (lane == 0) median_to_first_lds(D, f, l);
            team_sync<false>();
            const int cut = partition_lds(D, f, l, wBL, wBR, wPL, wSX, wacc, Lp, Rp, lane);
            if (cut < 0) break;
            if (l - cut > kSortThreshold) {
                if (lane == 0) { stack[2 * sp] = cut; stack[2 * sp + 1] = l | (depth << 24); }
                ++sp;
            }
            l = cut;
        }
        if (sp == 0) break;
        --sp;
        team_sync<false>();
        f = stack[2 * sp];
        l = stack[2 * sp + 1] & 0xffffff;
        depth = stack[2 * sp + 1] >> 24;
    }
    team_sync<false>();
}

#ifdef LF_SEED_STAMPS
__device__ int g_dbg_big[8], g_dbg_small[8];
__device__ long long g_dbg_t[8][4];
#endif
__device__ __forceinline__ void introsort_loop_wg(uint32_t* E, int n, unsigned long long* scratch, uint32_t* lds, int rows_cap)
{
    // LDS carve-up: phase 1 = the row tables of the global partitions; phase 2 = one private block per working wave (aliased)
    lds_u64* BL = as_lds<lds_u64>(lds);
    lds_u64* BR = BL + rows_cap;
    lds_i32* PL = (lds_i32*)(BR + rows_cap);
    lds_i32* SX = PL + rows_cap;                          // rows_cap + 1 entries
    unsigned long long* small_list = scratch;                          // ranges of <= kSmall elements cut straight from a global partition
    unsigned long long* block_list = scratch + (n / 16 + 64);          // ranges of kSmall < size <= kBlock: phase 1b
    uint32_t* Lpos = reinterpret_cast<uint32_t*>(block_list + (n / kSmall + 64));
    uint32_t* Rpos = Lpos + (n / 2 + 8);
    __shared__ int acc_[4];
    lds_i32* acc = as_lds<lds_i32>(acc_);
    __shared__ int big_stack[3 * 72];
    __shared__ int n_big, n_small, n_block, next_small, n_blk_small;
    __shared__ int cur[4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    auto pack = [](int f, int l, int depth) { return (unsigned long long)f | ((unsigned long long)l << 20) | ((unsigned long long)depth << 40); };
    if (t == 0) {
        n_big = 0; n_small = 0; n_block = 0; next_small = 0;
        const int depth0 = 2 * (31 - __clz(n));
        if (n > kBlock) { big_stack[0] = 0; big_stack[1] = n; big_stack[2] = depth0; n_big = 1; }
        else if (n > kSmall) block_list[n_block++] = pack(0, n, depth0);
        else if (n > kSortThreshold) small_list[n_small++] = pack(0, n, depth0);
    }
    __syncthreads();
    SEED_T(ta);
    // ---- phase 1: ranges of more than kBlock elements, in global memory
    for (;;) {
        if (t == 0) {
            if (n_big > 0) { --n_big; cur[0] = big_stack[3 * n_big]; cur[1] = big_stack[3 * n_big + 1]; cur[2] = big_stack[3 * n_big + 2]; cur[3] = 1; }
            else cur[3] = 0;
        }
        __syncthreads();
        if (!cur[3]) break;
        int f = cur[0], l = cur[1], depth = cur[2];
        __syncthreads();
        // the libstdc++ loop on this range: go on with the left part while it is big, park the right part
        while (l - f > kBlock) {
            if (depth == 0) {
                if (t == 0) heap_sort_range(E, f, l);
                __syncthreads();
                l = f;
                break;
            }
            --depth;
            if (t == 0) median_to_first(E, f, l);
            __syncthreads();
            const int cut = partition_global(E, f, l, BL, BR, PL, SX, acc, Lpos, Rpos, w, lane);
#ifdef LF_SEED_STAMPS
            if (t == 0) atomicAdd(&g_dbg_big[blockIdx.x % 8], 1);
#endif
            if (cut < 0) { l = f; break; }                           // no seed in the range: nothing to order
            if (t == 0) {
                const int rs = l - cut;
                if (rs > kBlock) { big_stack[3 * n_big] = cut; big_stack[3 * n_big + 1] = l; big_stack[3 * n_big + 2] = depth; ++n_big; }
                else if (rs > kSmall) block_list[n_block++] = pack(cut, l, depth);
                else if (rs > kSortThreshold) small_list[n_small++] = pack(cut, l, depth);
            }
            l = cut;
            __syncthreads();
        }
        if (t == 0) {
            if (l - f > kSmall) block_list[n_block++] = pack(f, l, depth);
            else if (l - f > kSortThreshold) small_list[n_small++] = pack(f, l, depth);
        }
        __syncthreads();
    }
    SEED_T(tb);
    // ---- phase 1b: every listed block is copied into LDS, partitioned there by the whole workgroup down to kSmall, its small
    // ranges are worked off in place by eight waves, and it is copied back
    {
        lds_u32* Dg = as_lds<lds_u32>(lds);
        uint32_t* xreg = lds + kBlock;
        lds_u16* cLp = as_lds<lds_u16>(xreg);
        lds_u16* cRp = cLp + kBlock / 2;
        lds_u64* bBL = as_lds<lds_u64>(lds + kBlock + kBlkX);
        lds_u64* bBR = bBL + kBlkRows;
        lds_i32* bPL = (lds_i32*)(bBR + kBlkRows);
        lds_i32* bSX = bPL + kBlkRows;
        lds_u32* blk_small = as_lds<lds_u32>(lds + kBlock + kBlkX + (kBlkRows * 6 + 8));          // pairs (f, l | depth << 24)
        const int nb = n_block;
        for (int bi = 0; bi < nb; ++bi) {
            const unsigned long long it = block_list[bi];
            const int gf = (int)(it & 0xfffffu), gl = (int)((it >> 20) & 0xfffffu);
            const int m = gl - gf;
            __syncthreads();
            if (t == 0) { acc[3] = 0; n_big = 0; n_blk_small = 0; next_small = 0; }
            __syncthreads();
            {
                int seeds = 0;
                for (int x0 = t; x0 < m; x0 += ST * kU) {
                    uint32_t v[kU];
#pragma unroll
                    for (int u = 0; u < kU; ++u) { const int x = x0 + u * ST; v[u] = x < m ? E[gf + x] : 0u; }
#pragma unroll
                    for (int u = 0; u < kU; ++u) { const int x = x0 + u * ST; if (x < m) { Dg[x] = v[u]; seeds |= (v[u] & 0xfffffu) != 0u; } }
                }
                if (__ballot(seeds) && lane == 0) lds_or(&acc[3], 1);
            }
            __syncthreads();
            if (acc[3] == 0) continue;                                  // no seed in the block
            if (t == 0) { big_stack[0] = 0; big_stack[1] = m; big_stack[2] = (int)(it >> 40); n_big = 1; }
            __syncthreads();
            for (;;) {
                if (t == 0) {
                    if (n_big > 0) { --n_big; cur[0] = big_stack[3 * n_big]; cur[1] = big_stack[3 * n_big + 1]; cur[2] = big_stack[3 * n_big + 2]; cur[3] = 1; }
                    else cur[3] = 0;
                }
                __syncthreads();
                if (!cur[3]) break;
                int f = cur[0], l = cur[1], depth = cur[2];
                __syncthreads();
                while (l - f > kSmall) {
                    if (depth == 0) {
                        if (t == 0) heap_sort_range(lds, f, l);
                        __syncthreads();
                        l = f;
                        break;
                    }
                    --depth;
                    if (t == 0) median_to_first_lds(Dg, f, l);
                    __syncthreads();
                    const int cut = partition_lds_coop(Dg, f, l, bBL, bBR, bPL, bSX, acc, cLp, cRp, w, lane);
                    if (cut < 0) { l = f; break; }
                    if (t == 0) {
                        const int rs = l - cut;
                        if (rs > kSmall) { big_stack[3 * n_big] = cut; big_stack[3 * n_big + 1] = l; big_stack[3 * n_big + 2] = depth; ++n_big; }
                        else if (rs > kSortThreshold) { blk_small[2 * n_blk_small] = (uint32_t)cut; blk_small[2 * n_blk_small + 1] = (uint32_t)l | ((uint32_t)depth << 24); ++n_blk_small; }
                    }
                    l = cut;
                    __syncthreads();
                }
                if (t == 0 && l - f > kSortThreshold) { blk_small[2 * n_blk_small] = (uint32_t)f; blk_small[2 * n_blk_small + 1] = (uint32_t)l | ((uint32_t)depth << 24); ++n_blk_small; }
                __syncthreads();
            }
            // the block's small ranges: eight waves, in place (their private lists alias the workgroup's place lists, which are idle now)
            if (w < SW2) {
                uint32_t* mine = xreg + (size_t)w * kBlkWaveWords;
                lds_u16* Lp = as_lds<lds_u16>(mine);
                lds_u16* Rp = Lp + kSmall / 2;
                lds_u64* wBL = as_lds<lds_u64>(mine + kSmall / 2);
                lds_u64* wBR = wBL + 18;
                lds_i32* wPL = (lds_i32*)(wBR + 18);
                lds_i32* wSX = wPL + 18;
                lds_i32* wacc = wSX + 20;
                lds_i32* stack = wacc + 8;
                const int total = n_blk_small;
                for (;;) {
                    int idx = 0;
                    if (lane == 0) idx = atomicAdd(&next_small, 1);
                    idx = __shfl(idx, 0);
                    if (idx >= total) break;
                    const uint32_t a = blk_small[2 * idx], b = blk_small[2 * idx + 1];
                    wave_subtree(Dg, lds, (int)a, (int)(b & 0xffffffu), (int)(b >> 24), wBL, wBR, wPL, wSX, wacc, Lp, Rp, stack, lane);
                }
            }
            __syncthreads();
            for (int x0 = t; x0 < m; x0 += ST * kU) {
#pragma unroll
                for (int u = 0; u < kU; ++u) { const int x = x0 + u * ST; if (x < m) E[gf + x] = Dg[x]; }
            }
        }
        __syncthreads();
        if (t == 0) next_small = 0;
        __syncthreads();
    }
    SEED_T(tc);
    // ---- phase 2: a wave per range cut straight from a global partition, copied into its private LDS block
    if (w < SW2) {
        uint32_t* mine = lds + (size_t)w * kWaveWords;
        lds_u32* D = as_lds<lds_u32>(mine);
        lds_u16* Lp = (lds_u16*)(D + kSmall);
        lds_u16* Rp = Lp + kSmall / 2;
        lds_u64* wBL = as_lds<lds_u64>(mine + kSmall + kSmall / 2);
        lds_u64* wBR = wBL + 18;
        lds_i32* wPL = (lds_i32*)(wBR + 18);
        lds_i32* wSX = wPL + 18;
        lds_i32* wacc = wSX + 20;
        lds_i32* stack = wacc + 8;                                // 64 x (f, l | depth << 24)
        const int total = n_small;
#ifdef LF_SEED_STAMPS
        if (t == 0) g_dbg_small[blockIdx.x % 8] = total + 1000 * n_block;
#endif
        for (;;) {
            int idx = 0;
            if (lane == 0) idx = atomicAdd(&next_small, 1);
            idx = __shfl(idx, 0);
            if (idx >= total) break;
            const unsigned long long it = small_list[idx];
            const int gf = (int)(it & 0xfffffu), gl = (int)((it >> 20) & 0xfffffu);
            const int m = gl - gf;
            int seeds = 0;
            for (int x = lane; x < m; x += 64) { const uint32_t v = E[gf + x]; D[x] = v; seeds |= (v & 0xfffffu) != 0u; }
            if (!__ballot(seeds)) continue;                           // no seed: leave it as it is
            team_sync<false>();
            wave_subtree(D, mine, 0, m, (int)(it >> 40), wBL, wBR, wPL, wSX, wacc, Lp, Rp, stack, lane);
            for (int x = lane; x < m; x += 64) E[gf + x] = D[x];
        }
    }
    __syncthreads();
#ifdef LF_SEED_STAMPS
    { const long long td = (long long)wall_clock64(); if (t == 0) { g_dbg_t[blockIdx.x % 8][0] = tb - ta; g_dbg_t[blockIdx.x % 8][1] = tc - tb; g_dbg_t[blockIdx.x % 8][2] = td - tc; } }
#endif
}

// one stable 4-bit counting pass (same scheme as k_lsd_order.hip's radix_pass: every thread owns a contiguous run; [16][1024] counters)
__device__ __forceinline__ void seed_radix_pass(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int n, int shift,
                                                uint32_t* cnt /*[SNB][ST]*/, int* tot, int* base)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int seg = (n + ST - 1) / ST;
    const int i0 = min(n, t * seg), i1 = min(n, i0 + seg);
    for (int b = 0; b < SNB; ++b) cnt[b * ST + t] = 0;
    for (int i = i0; i < i1; ++i) cnt[(int)((src[i] >> shift) & (SNB - 1)) * ST + t]++;
    __syncthreads();
    for (int bb = 0; bb < SNB / SW; ++bb) {
        const int b = wave * (SNB / SW) + bb;
        int carry = 0;
        for (int c = 0; c < ST / 64; ++c) {
            const int v = (int)cnt[b * ST + c * 64 + lane];
            const int inc = wave_incl_scan_i(v, lane);
            cnt[b * ST + c * 64 + lane] = (uint32_t)(carry + inc - v);
            carry += __shfl(inc, 63);
        }
        if (lane == 0) tot[b] = carry;
    }
    __syncthreads();
    if (wave == 0) {
        const int v = lane < SNB ? tot[lane] : 0;
        const int inc = wave_incl_scan_i(v, lane);
        if (lane < SNB) base[lane] = inc - v;
    }
    __syncthreads();
    for (int i = i0; i < i1; ++i) {
        const uint32_t it = src[i];
        const int b = (int)((it >> shift) & (SNB - 1));
        dst[(uint32_t)base[b] + cnt[b * ST + t]++] = it;
    }
    __syncthreads();
}

__global__ __launch_bounds__(ST) void k_lsd_seed32(LsdParams p, const int* __restrict__ n_rec, const unsigned long long* __restrict__ maxgrad,
                                                   const uint32_t* __restrict__ c_xy, const double* __restrict__ c_mod,
                                                   const uint32_t* __restrict__ l_addr, const double* __restrict__ l_mod, const int* __restrict__ n_low,
                                                   unsigned long long* __restrict__ sort_a, unsigned long long* __restrict__ sort_b,
                                                   uint32_t* __restrict__ order_a, uint32_t* __restrict__ order_b, int rows_cap)
{
    extern __shared__ uint32_t seed_lds[];
    __shared__ int tot[SNB];
    __shared__ int base[SNB];
    const int pc = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const size_t Ps = (size_t)p.Hs * p.Ws;
    const size_t o = (size_t)pc * Ps;
    const int nd = n_rec[pc];
    if (nd == 0) return;
    uint32_t* E = reinterpret_cast<uint32_t*>(sort_a + o);
    unsigned long long* small_list = sort_b + o;
    uint32_t* A = order_a + o;
    uint32_t* B = order_b + o;
    const int Wg = p.Ws - 1, Hg = p.Hs - 1;
    const int n = Wg * Hg;
    const double max_grad = __longlong_as_double((long long)maxgrad[pc]);
    const double bin_coef = (max_grad > 0) ? (double)(p.n_bins - 1) / max_grad : 0;
    SEED_T(t0);
    // ---- phase 0
    for (int i = t; i < n; i += ST) E[i] = 0u;
    __syncthreads();
    for (int e = t; e < nd; e += ST) {
        const uint32_t xy = c_xy[o + e];
        const int bin = (int)(c_mod[o + e] * bin_coef);
        E[(int)(xy >> 16) * Wg + (int)(xy & 0xffffu)] = ((uint32_t)bin << 20) | (uint32_t)(e + 1);
    }
    const int nl = n_low[pc];
    for (int j = t; j < nl; j += ST) {
        const uint32_t a = l_addr[o + j];
        const int y = (int)(a / (uint32_t)p.Ws), x = (int)(a - (uint32_t)y * (uint32_t)p.Ws);
        E[y * Wg + x] = (uint32_t)(int)(l_mod[o + j] * bin_coef) << 20;
    }
    __syncthreads();
    SEED_T(t1);
    // ---- phases 1, 2
    introsort_loop_wg(E, n, small_list, seed_lds, rows_cap);          // (the scratch: this problem's slice of sort_b, Ps >= 3 n / 4 + 64 entries)
    SEED_T(t2);
    // ---- phase 3: the seeds in array order ...
    int* rowc = reinterpret_cast<int*>(seed_lds);                  // [rows + 1]
    const int R = (n + 63) >> 6;
    for (int r0 = w * kU; r0 < R; r0 += SW * kU) {
        uint32_t v[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) { const int i = (r0 + u) * 64 + lane; v[u] = i < n ? E[i] : 0u; }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const unsigned long long b = __ballot((v[u] & 0xfffffu) != 0u);
            if (lane == 0 && r0 + u < R) rowc[r0 + u] = __popcll(b);
        }
    }
    __syncthreads();
    if (w == 0) {
        int carry = 0;
        for (int r0 = 0; r0 < R; r0 += 64) {
            const int r = r0 + lane;
            const int c = r < R ? rowc[r] : 0;
            const int inc = wave_incl_scan_i(c, lane);
            if (r < R) rowc[r] = carry + inc - c;
            carry += __shfl(inc, 63);
        }
    }
    __syncthreads();
    for (int r0 = w * kU; r0 < R; r0 += SW * kU) {
        uint32_t v[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) { const int i = (r0 + u) * 64 + lane; v[u] = i < n ? E[i] : 0u; }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const bool seed = (v[u] & 0xfffffu) != 0u;
            const unsigned long long b = __ballot(seed);
            if (seed) B[rowc[r0 + u] + __popcll(b & ((1ull << lane) - 1ull))] = ((uint32_t)((p.n_bins - 1) - (int)key_of(v[u])) << 20) | ((v[u] & 0xfffffu) - 1u);
        }
    }
    __syncthreads();
    // ... and the final insertion sort: stable by bin, highest bin first (n_seeds == nd: every defined pixel is in the array once)
    uint32_t* cnt = seed_lds;
    seed_radix_pass(B, A, nd, 20, cnt, tot, base);
    seed_radix_pass(A, B, nd, 24, cnt, tot, base);
    seed_radix_pass(B, A, nd, 28, cnt, tot, base);
#ifdef LF_SEED_STAMPS
    { const long long t3 = (long long)wall_clock64(); if (t == 0 && pc < 6) printf("[seed32] problem %d: n %d seeds %d low %d | phase0 %lld  loop %lld (%d big partitions, %d small ranges)  phase3 %lld  | global %lld  blocks %lld  small %lld (x10 ns)\n", pc, n, nd, nl, t1 - t0, t2 - t1, g_dbg_big[pc % 8], g_dbg_small[pc % 8], t3 - t2, g_dbg_t[pc % 8][0], g_dbg_t[pc % 8][1], g_dbg_t[pc % 8][2]); }
#endif
}

// debug / test entry: std::sort(compare_norm) of n elements (key << 20 | index + 1) given in E; leaves E as the introsort loop
// + final insertion sort leave it (the insertion sort as stable passes).  One workgroup.
__global__ __launch_bounds__(ST) void k_std_sort_debug(uint32_t* __restrict__ E, uint32_t* __restrict__ tmp, unsigned long long* __restrict__ small_list, int n, int rows_cap)
{
    extern __shared__ uint32_t seed_lds[];
    __shared__ int tot[SNB];
    __shared__ int base[SNB];
    introsort_loop_wg(E, n, small_list, seed_lds, rows_cap);
    // stable by key DESCENDING: passes over (1023 - key)
    for (int i = threadIdx.x; i < n; i += ST) { const uint32_t v = E[i]; E[i] = ((1023u - key_of(v)) << 20) | (v & 0xfffffu); }
    __syncthreads();
    uint32_t* cnt = seed_lds;
    seed_radix_pass(E, tmp, n, 20, cnt, tot, base);
    seed_radix_pass(tmp, E, n, 24, cnt, tot, base);
    seed_radix_pass(E, tmp, n, 28, cnt, tot, base);
    for (int i = threadIdx.x; i < n; i += ST) { const uint32_t v = tmp[i]; E[i] = ((1023u - key_of(v)) << 20) | (v & 0xfffffu); }
}

// rows of 64 elements the row tables must hold for an n-element array (a multiple of 64, so that the tables stay 8-byte aligned)
static int seed_rows_cap(long long n) { return (int)(((n + 63) / 64 + 1 + 63) / 64 * 64); }
static size_t seed_lds_bytes(int rows_cap)
{
    size_t words = (size_t)rows_cap * 6 + 128;
    if (words < (size_t)SW2 * kWaveWords) words = (size_t)SW2 * kWaveWords;
    if (words < (size_t)kBlkWords) words = (size_t)kBlkWords;
    if (words < (size_t)SNB * ST) words = (size_t)SNB * ST;
    return words * sizeof(uint32_t);
}

bool lsd_seed32_supported(const LsdParams& p)
{
    const long long n = (long long)(p.Hs - 1) * (p.Ws - 1);
    return p.n_bins <= 1024 && (long long)p.Hs * p.Ws < (1 << 20) && n >= 1 && seed_lds_bytes(seed_rows_cap(n)) <= (size_t)kMaxLdsBytes;
}

void launch_lsd_seed32(const LsdParams& p, int n_frames, const int* n_rec, const unsigned long long* maxgrad, const uint32_t* c_xy,
                       const double* c_mod, const uint32_t* l_addr, const double* l_mod, const int* n_low,
                       unsigned long long* sort_a, unsigned long long* sort_b, uint32_t* order_a, uint32_t* order_b, hipStream_t s)
{
    const int rows_cap = seed_rows_cap((long long)(p.Hs - 1) * (p.Ws - 1));
    const size_t lds = seed_lds_bytes(rows_cap);
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_lsd_seed32), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_lsd_seed32, dim3(n_frames * 3), dim3(ST), lds, s, p, n_rec, maxgrad, c_xy, c_mod, l_addr, l_mod, n_low,
                       sort_a, sort_b, order_a, order_b, rows_cap);
}

// n < 2^20 and at most kMaxLdsBytes of row tables: false otherwise
bool launch_std_sort_debug(uint32_t* E, uint32_t* tmp, unsigned long long* small_list, int n, hipStream_t s)
{
    const int rows_cap = seed_rows_cap(n);
    const size_t lds = seed_lds_bytes(rows_cap);
    if (n < 1 || n >= (1 << 20) || lds > (size_t)kMaxLdsBytes) return false;
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_std_sort_debug), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_std_sort_debug, dim3(1), dim3(ST), lds, s, E, tmp, small_list, n, rows_cap);
    return true;
}

}  // namespace lf
