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
// L element or at R_K, which now holds one).
//
// Round 5: THE SPARSE FORM.  The gradient image is flat almost everywhere (bin 0: ~95 % of the pixels of a lane frame), pixels of
// bin 0 never seed a region, and the sort's decisions depend on keys alone -- so the zeros need not exist.  A problem is its
// EXPLICIT LIST: the pixels with a non-zero bin (the defined ones from the compact arrays of k_lsd_order, the undefined ones
// from k_lsd_grad's "low" records), (position, key << 20 | seed + 1) sorted by position; every other position of the array is an
// implicit, anonymous zero.  The top of the introsort loop then is a CHAIN of steps on ONE sparse range [f, l):
//   fold    the pivot is a zero (the usual case: the median of three mostly flat pixels).  L = the zeros, R = everything: the
//           k-th zero from the left takes the k-th element from the right, K = the largest k with zeros(lo .. hi-k-1) >= k, and
//           everything right of the cut is flat -- it cannot be observed and is dropped.  For the list: the entries of the last K
//           positions move to L_k = lo + k - 1 + (entries in front of the k-th zero), the others stay; the new list is the MERGE
//           of the two (ranks by binary search: one search per entry), O(entries log entries) whatever the range's length.
//   split   the pivot has key kp > 0.  L = zeros + entries with key <= kp, R = entries with key >= kp, "above" = entries with
//           key > kp.  L_k = the k-th position that holds no entry above the pivot; the swaps touch at most |R| entries.  The
//           left part [f, cut) holds no zero at all -- it is written out as a DENSE range -- the right part stays sparse.
// The chain ends when the range is short or mostly explicit (it is then written out densely, zeros included) or holds no seed
// (dropped).  What it leaves is a dense array E' of about as many elements as the list had (4 - 8 k for a lane frame instead
// of 130 k) in array order, as ranges with their remaining depth allowance; the introsort loop below works them off as before
// (workgroup partitions in global memory / in an LDS block, single waves on ranges of <= 1024 elements), and the final
// insertion sort is a stable counting sort by bin of the seeds in E' order.  tools/probe/seed_sparse_proto.cpp is the same chain
// written for the host and checked against the real std::sort (3 000 random arrays + the bins of lane / clutter / camera frames).
//
// One workgroup per problem (frame, colour).  Elements are u32: bin << 20 | payload (compact index + 1 of a pixel with a
// defined gradient, 0 for the others: only seeds need to be told apart).  A RANGE WITHOUT A SEED IS NEVER PARTITIONED: the loop
// only permutes a range within itself, so what it does to a range that holds no seed cannot be seen in the result.
// The heap sort of a range that used up the depth limit (never seen on image data; tested with adversarial keys through
// lf_debug_std_sort) is libstdc++'s __heap_select + __sort_heap replayed by one lane.
#include <cstdlib>
#include "common.h"
#include "lsd_bitplane.h"

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
// partitions are chains of dependent round trips, not throughput).
#ifndef LF_SEED_THREADS
#define LF_SEED_THREADS 256
#endif
#ifndef LF_SEED_WAVES2
#define LF_SEED_WAVES2 4
#endif
#ifndef LF_SEED_BLOCK
#define LF_SEED_BLOCK 4096
#endif
#ifndef LF_SEED_EPT
#define LF_SEED_EPT 32
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
constexpr int kEPT = LF_SEED_EPT;    // list entries per thread of the LDS form of the chain
constexpr int kListLds = kEPT * ST;  // the explicit list's positions stay in LDS up to this many entries
constexpr int kChainDense = 1024;    // a sparse range this short is written out densely
constexpr int kMaxRanges = 48;       // dense ranges the chain can leave (one per split + the last: the depth allowance is 2 lg n <= 42)
constexpr int kRowsLds = 512;        // row tables in LDS for dense ranges of up to 64 x this many elements (longer ones: tables in global memory)
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
// the same three on either kind of table (the row tables of an oversized range live in global memory)
__device__ __forceinline__ void tab_add(lds_i32* p, int v) { lds_add(p, v); }
__device__ __forceinline__ void tab_min(lds_i32* p, int v) { lds_min(p, v); }

__device__ __forceinline__ int wave_incl_scan_i(int v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int n = __shfl_up(v, d);
        if (lane >= d) v += n;
    }
    return v;
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

// position of the t-th (1-based) set bit of b counted from bit 63 downwards: the largest p with popcount(b >> p) >= t
__device__ __forceinline__ int select_from_top(unsigned long long b, int t)
{
    int pos = 0;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1)
        if (__popcll(b >> (pos + s)) >= t) pos += s;
    return pos;
}

// Passes 2 - 4 and the cut, given the row ballots of [lo, hi) in BL / BR (pass 1 differs between the users).  EP: the
// elements (global or LDS), LP: the place lists (u32 in global scratch / u16 in LDS), positions relative to `org`; T64 / T32: the
// row tables (LDS, or global memory for a range beyond the LDS tables).
template <bool COOP, typename EP, typename LP, typename T64, typename T32>
__device__ __forceinline__ int partition_tail(EP E, int org, int lo, int hi, T64 BL, T64 BR, T32 PL, T32 SX,
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
template <typename T64, typename T32>
__device__ __forceinline__ int partition_global(uint32_t* E, int f, int l, T64 BL, T64 BR, T32 PL, T32 SX,
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
__device__ int g_dbg_big[8], g_dbg_small[8], g_dbg_chain[8][4];
__device__ long long g_dbg_t[8][4];
#endif

// a dense range the chain left: elements E[off, off + len), depth allowance left
struct SeedRange { int off, len, depth; };

// The introsort loop over the listed ranges of E (n = their total extent): the phases 1, 1b, 2 of the dense form.  scratch: global,
// 3 n / 4 + 192 u64 entries (the list of small ranges, the list of blocks, then the two place lists of the global partitions);
// gtab: global row tables (6 words per 64 elements of the longest range + 16) for ranges beyond rows_cap rows.
__device__ __forceinline__ void introsort_loop_wg(uint32_t* E, int n, const SeedRange* ranges, int n_ranges, unsigned long long* scratch,
                                                  uint32_t* lds, int rows_cap, uint32_t* gtab)
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
    __shared__ int big_stack[3 * (72 + kMaxRanges)];
    __shared__ int n_big, n_small, n_block, next_small, n_blk_small;
    __shared__ int cur[4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    auto pack = [](int f, int l, int depth) { return (unsigned long long)f | ((unsigned long long)l << 24) | ((unsigned long long)depth << 48); };
    if (t == 0) {
        n_big = 0; n_small = 0; n_block = 0; next_small = 0;
        for (int i = 0; i < n_ranges; ++i) {
            const int f = ranges[i].off, l = f + ranges[i].len, d = ranges[i].depth, sz = ranges[i].len;
            if (sz > kBlock) { big_stack[3 * n_big] = f; big_stack[3 * n_big + 1] = l; big_stack[3 * n_big + 2] = d; ++n_big; }
            else if (sz > kSmall) block_list[n_block++] = pack(f, l, d);
            else if (sz > kSortThreshold) small_list[n_small++] = pack(f, l, d);
        }
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
            int cut;
            if (((l - f - 1 + 63) >> 6) <= rows_cap - 1) cut = partition_global(E, f, l, BL, BR, PL, SX, acc, Lpos, Rpos, w, lane);
            else {
                const int rc = ((l - f) >> 6) + 2;
                unsigned long long* gBL = reinterpret_cast<unsigned long long*>(gtab);
                unsigned long long* gBR = gBL + rc;
                int* gPL = reinterpret_cast<int*>(gBR + rc);
                int* gSX = gPL + rc;
                cut = partition_global(E, f, l, gBL, gBR, gPL, gSX, acc, Lpos, Rpos, w, lane);
            }
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
    // ranges are worked off in place by the waves, and it is copied back
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
            const int gf = (int)(it & 0xffffffu), gl = (int)((it >> 24) & 0xffffffu);
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
            if (t == 0) { big_stack[0] = 0; big_stack[1] = m; big_stack[2] = (int)(it >> 48); n_big = 1; }
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
            // the block's small ranges: the waves, in place (their private lists alias the workgroup's place lists, which are idle now)
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
            const int gf = (int)(it & 0xffffffu), gl = (int)((it >> 24) & 0xffffffu);
            const int m = gl - gf;
            int seeds = 0;
            for (int x = lane; x < m; x += 64) { const uint32_t v = E[gf + x]; D[x] = v; seeds |= (v & 0xfffffu) != 0u; }
            if (!__ballot(seeds)) continue;                           // no seed: leave it as it is
            team_sync<false>();
            wave_subtree(D, mine, 0, m, (int)(it >> 48), wBL, wBR, wPL, wSX, wacc, Lp, Rp, stack, lane);
            for (int x = lane; x < m; x += 64) E[gf + x] = D[x];
        }
    }
    __syncthreads();
#ifdef LF_SEED_STAMPS
    { const long long td = (long long)wall_clock64(); if (t == 0) { g_dbg_t[blockIdx.x % 8][0] = tb - ta; g_dbg_t[blockIdx.x % 8][1] = tc - tb; g_dbg_t[blockIdx.x % 8][2] = td - tc; } }
#endif
}

// ============================================================================================================================
// The chain: the top of the introsort loop on the explicit list (header).  One problem's global work areas, each of at least
// `cap` u32 words (cap >= the array length n and >= 1024):
struct SeedWork {
    uint32_t* E;              // the dense array the chain writes out
    uint32_t* V[2];           // the list's values (bin << 20 | seed + 1), double buffered
    uint32_t* P[2];           // the list's positions when it does not fit LDS (double buffered); P[0] also stages a split's new list for the LDS form
    uint32_t* PP;             // split: the positions of the entries above the pivot; the dense phase's global row tables; the final passes' second buffer
    uint32_t* T;              // split: the entry at the k-th L place (or none)
    uint32_t* RT;             // row tables of a list that does not fit LDS (6 words per 64 entries)
    unsigned long long* dscratch;   // the dense phase's range lists and place lists: aliases V[1] | P[0]
    uint32_t* out;            // the seeds in their final order (aliases P[1])
};

// the chain's state: LDS, written by single threads between barriers
struct ChainState {
    int f, l, depth, a, b, seeds, cur, e_used, n_ranges, stop;
    int md_idx[4];
    uint32_t md_val[4];
    int K, eK, cut, nP, nG, minR, lseeds, nright, wtot[SW];
    SeedRange ranges[kMaxRanges];
};

// first index in [a, b) whose position is >= q
template <typename PT>
__device__ __forceinline__ int list_lower_bound(PT P, int a, int b, int q)
{
    while (a < b) {
        const int mid = (a + b) >> 1;
        if ((int)P[mid] < q) a = mid + 1; else b = mid;
    }
    return a;
}

// exclusive scan of the R row counts in cnt (LDS or global) by one wave; returns the total in every lane
template <typename CT>
__device__ __forceinline__ int wave_scan_rows(CT cnt, int R, int lane)
{
    int carry = 0;
    for (int r0 = 0; r0 < R; r0 += 64) {
        const int r = r0 + lane;
        const int c = r < R ? (int)cnt[r] : 0;
        const int inc = wave_incl_scan_i(c, lane);
        if (r < R) cnt[r] = carry + inc - c;
        carry += __shfl(inc, 63);
    }
    return carry;
}

// [off, off + len) of E <- the sparse range [f, l) with the list entries [a, b)
template <typename PT>
__device__ __forceinline__ void chain_materialize(uint32_t* E, int off, int f, int l, PT P, const uint32_t* V, int a, int b)
{
    const int t = threadIdx.x;
    for (int i = t; i < l - f; i += ST) E[off + i] = 0u;
    __syncthreads();
    for (int i = a + t; i < b; i += ST) { const uint32_t v = V[i]; if (v) E[off + (int)P[i] - f] = v; }
    __syncthreads();
}

// PL: the positions in LDS (single buffer Pl, new places through registers), else in W.P[cur] (double buffered)
template <bool PL>
__device__ __forceinline__ void sparse_chain(const SeedWork& W, lds_u32* Pl, lds_u32* tab, ChainState* cs_generic, int n, int m0, int nseeds)
{
    typedef __attribute__((address_space(3))) ChainState lds_cs;
    lds_cs* cs = (lds_cs*)(__attribute__((address_space(3))) void*)cs_generic;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) {
        cs->f = 0; cs->l = n; cs->depth = 2 * (31 - __clz(n)); cs->a = 0; cs->b = m0; cs->seeds = nseeds; cs->cur = 0; cs->e_used = 0; cs->n_ranges = 0;
    }
    __syncthreads();
    for (;;) {
        const int f = cs->f, l = cs->l, a = cs->a, b = cs->b, depth = cs->depth, cur = cs->cur, seeds = cs->seeds;
        const int len = l - f, m = b - a;
        uint32_t* V = W.V[cur];
        uint32_t* V2 = W.V[cur ^ 1];
        uint32_t* Pg = W.P[cur];
        uint32_t* Pg2 = W.P[cur ^ 1];
        __syncthreads();                                                  // everybody has read the state
        if (seeds == 0) break;
        if (len <= kChainDense || (len - m) * 4 <= len || depth == 0 || cs->n_ranges >= kMaxRanges - 1) {
            const int off = cs->e_used;
            if (PL) chain_materialize(W.E, off, f, l, Pl, V, a, b); else chain_materialize(W.E, off, f, l, Pg, V, a, b);
            if (t == 0) { const int k = cs->n_ranges; cs->ranges[k].off = off; cs->ranges[k].len = len; cs->ranges[k].depth = depth; cs->n_ranges = k + 1; cs->e_used = off + len; }
            __syncthreads();
            break;
        }
        // ---- the median of three: the values at f, f + 1, mid, l - 1
        if (t < 4) {
            const int q = t == 0 ? f : (t == 1 ? f + 1 : (t == 2 ? f + len / 2 : l - 1));
            const int i = PL ? list_lower_bound(Pl, a, b, q) : list_lower_bound(Pg, a, b, q);
            const bool hit = i < b && (int)(PL ? Pl[i] : Pg[i]) == q;
            cs->md_idx[t] = hit ? i : -1;
            cs->md_val[t] = hit ? V[i] : 0u;
        }
        __syncthreads();
        const uint32_t vf = cs->md_val[0], va = cs->md_val[1], vb = cs->md_val[2], vc = cs->md_val[3];
        const int xf = cs->md_idx[0];
        int pk;                                                            // 1: f + 1, 2: mid, 3: l - 1
        if (comp(va, vb)) pk = comp(vb, vc) ? 2 : (comp(va, vc) ? 3 : 1);
        else pk = comp(va, vc) ? 1 : (comp(vb, vc) ? 3 : 2);
        const uint32_t pv = cs->md_val[pk];
        const int xp = cs->md_idx[pk];
        const int pick = pk == 1 ? f + 1 : (pk == 2 ? f + len / 2 : l - 1);
        __syncthreads();                                                  // md_* are free again
        int a1 = a;
        if (xp >= 0) {                                                     // the pivot is an entry: its slot takes what was at f (nothing: the entry is dead, value 0)
            if (t == 0) V[xp] = vf;
            if (xf >= 0) a1 = a + 1;
            __syncthreads();
        } else if (vf != 0u) {                                             // the pivot is a zero and f held an entry: that entry moves to `pick`
            const int ins = PL ? list_lower_bound(Pl, a, b, pick) : list_lower_bound(Pg, a, b, pick);
            for (int base = a + 1; base < ins; base += ST) {               // entries [a + 1, ins) one place down
                const int i = base + t;
                const bool on = i < ins;
                uint32_t p_ = 0, v_ = 0;
                if (on) { p_ = PL ? Pl[i] : Pg[i]; v_ = V[i]; }
                __syncthreads();
                if (on) { if (PL) Pl[i - 1] = p_; else Pg[i - 1] = p_; V[i - 1] = v_; }
                __syncthreads();
            }
            if (t == 0) { if (PL) Pl[ins - 1] = (uint32_t)pick; else Pg[ins - 1] = (uint32_t)pick; V[ins - 1] = vf; }
            __syncthreads();
        }
        const int lo = f + 1, hi = l;
        const int mm = b - a1;
        if (key_of(pv) == 0u) {
            // ================================================ FOLD
            const int Zt = (hi - lo) - mm;
            // K: with e entries inside the last k positions, k is feasible iff 2 k - e <= Zt; the entry count steps at k = hi - position
            for (int e = t; e <= mm; e += ST) {
                const int ks = e == 0 ? 0 : hi - (int)(PL ? Pl[b - e] : Pg[b - e]);
                const int ksn = e < mm ? hi - (int)(PL ? Pl[b - e - 1] : Pg[b - e - 1]) : 0;
                const bool feas = 2 * ks - e <= Zt;
                const bool feasn = e < mm && 2 * ksn - (e + 1) <= Zt;
                if (feas && !feasn) {
                    int kmax = (Zt + e) / 2;
                    if (e < mm) kmax = min(kmax, ksn - 1);
                    kmax = min(kmax, hi - lo);
                    cs->K = kmax; cs->eK = e;
                }
            }
            __syncthreads();
            const int K = cs->K, eK = cs->eK;
            const int ns = mm - eK;                                         // the entries that stay: [a1, a1 + ns)
            // the cut (one thread, while the others place the entries): K == 0 ? L_1 : min(L_{K+1}, R_K), L over the list as it is
            if (t == ST - 1) {
                auto LkFull = [&](int k) {
                    int x = 0, y = mm;
                    while (x < y) { const int mid = (x + y) >> 1; const int g = (int)(PL ? Pl[a1 + mid] : Pg[a1 + mid]) - lo - mid; if (g < k) x = mid + 1; else y = mid; }
                    return lo + (k - 1) + x;
                };
                int cut;
                if (K == 0) cut = LkFull(1);
                else { cut = hi - K; if (Zt >= K + 1) cut = min(cut, LkFull(K + 1)); }
                cs->cut = cut;
            }
            // new places: a moved entry (the e-th from the right, k = hi - position) goes to L_k = lo + k - 1 + c, c = the staying
            // entries with fewer than k zeros in front of them; a staying entry keeps its position and moves up by the moved
            // entries that land in front of it (those with k <= its zeros)
            auto place = [&](int i, int& ni, uint32_t& np) {
                const int pos = (int)(PL ? Pl[i] : Pg[i]);
                if (i >= b - eK) {
                    const int e = b - i, k = hi - pos;
                    int x = 0, y = ns;
                    while (x < y) { const int mid = (x + y) >> 1; const int g = (int)(PL ? Pl[a1 + mid] : Pg[a1 + mid]) - lo - mid; if (g < k) x = mid + 1; else y = mid; }
                    ni = x + e - 1;
                    np = (uint32_t)(lo + (k - 1) + x);
                } else {
                    const int j = i - a1;
                    const int z = min(pos - lo - j, K);
                    const int first = PL ? list_lower_bound(Pl, a1, b, hi - z) : list_lower_bound(Pg, a1, b, hi - z);
                    ni = j + (b - first);
                    np = (uint32_t)pos;
                }
            };
            if (PL) {
                uint32_t np[kEPT];
                int ni[kEPT];
#pragma unroll
                for (int u = 0; u < kEPT; ++u) {
                    const int i = a1 + t + u * ST;
                    ni[u] = -1;
                    if (i < b) { place(i, ni[u], np[u]); V2[a1 + ni[u]] = V[i]; }
                }
                __syncthreads();
#pragma unroll
                for (int u = 0; u < kEPT; ++u) if (ni[u] >= 0) Pl[a1 + ni[u]] = np[u];
            } else {
                for (int i = a1 + t; i < b; i += ST) {
                    int ni; uint32_t np;
                    place(i, ni, np);
                    Pg2[a1 + ni] = np;
                    V2[a1 + ni] = V[i];
                }
            }
            __syncthreads();
            if (t == 0) { cs->l = cs->cut; cs->depth = depth - 1; cs->a = a1; cs->cur = cur ^ 1; }
            __syncthreads();
#ifdef LF_SEED_STAMPS
            if (t == 0) atomicAdd(&g_dbg_chain[blockIdx.x % 8][0], 1);
#endif
        } else {
            // ================================================ SPLIT
            const uint32_t kp = key_of(pv);
            const int R = (mm + 63) >> 6;
            // row tables: ballots of the entries above the pivot (key > kp) and of the R entries (key >= kp), their counts
            lds_u64* lBA = (lds_u64*)tab;
            lds_u64* lBG = lBA + kListLds / 64;
            lds_i32* lCA = (lds_i32*)(lBG + kListLds / 64);
            lds_i32* lCG = lCA + kListLds / 64;
            const int rc = R + 2;
            unsigned long long* gBA = reinterpret_cast<unsigned long long*>(W.RT);
            unsigned long long* gBG = gBA + rc;
            int* gCA = reinterpret_cast<int*>(gBG + rc);
            int* gCG = gCA + rc;
#define TBA(r) (PL ? lBA[r] : gBA[r])
#define TBG(r) (PL ? lBG[r] : gBG[r])
#define TCA(r) (PL ? lCA[r] : gCA[r])
#define TCG(r) (PL ? lCG[r] : gCG[r])
            const unsigned long long ltm = (1ull << lane) - 1ull;
            for (int r = w; r < R; r += SW) {
                const int i = a1 + r * 64 + lane;
                const uint32_t v = i < b ? V[i] : 0u;
                const unsigned long long ba = __ballot(v != 0u && key_of(v) > kp);
                const unsigned long long bg = __ballot(v != 0u && key_of(v) >= kp);
                if (lane == 0) {
                    if (PL) { lBA[r] = ba; lBG[r] = bg; lCA[r] = __popcll(ba); lCG[r] = __popcll(bg); }
                    else { gBA[r] = ba; gBG[r] = bg; gCA[r] = __popcll(ba); gCG[r] = __popcll(bg); }
                }
            }
            if (t == 0) { cs->K = 0; cs->minR = 0x7fffffff; cs->lseeds = (pv & 0xfffffu) != 0u; }
            __syncthreads();
            if (w == 0) { const int tot = PL ? wave_scan_rows(lCA, R, lane) : wave_scan_rows(gCA, R, lane); if (lane == 0) cs->nP = tot; }
            if (w == 1) { const int tot = PL ? wave_scan_rows(lCG, R, lane) : wave_scan_rows(gCG, R, lane); if (lane == 0) cs->nG = tot; }
            __syncthreads();
            const int nP = cs->nP, nG = cs->nG;
            // the positions above the pivot, compacted; T <- none
            for (int r = w; r < R; r += SW) {
                const int i = a1 + r * 64 + lane;
                const unsigned long long ba = TBA(r);
                if ((ba >> lane) & 1ull) W.PP[TCA(r) + __popcll(ba & ltm)] = PL ? Pl[i] : Pg[i];
            }
            for (int k = t; k <= nG; k += ST) W.T[k] = 0xffffffffu;
            __syncthreads();
            // L_k: the k-th position of [lo, hi) without an entry above the pivot
            const int nonA = (hi - lo) - nP;
            auto Lk = [&](int k) {
                if (k > nonA) return 0x7fffffff;
                int x = 0, y = nP;
                while (x < y) { const int mid = (x + y) >> 1; const int g = (int)W.PP[mid] - lo - mid; if (g < k) x = mid + 1; else y = mid; }
                return lo + (k - 1) + x;
            };
            // K = the R entries (rank k from the right) with L_k in front of them; R_K = the leftmost of them
            {
                int wk = 0, wmin = 0x7fffffff;
                for (int r = w; r < R; r += SW) {
                    const int i = a1 + r * 64 + lane;
                    const unsigned long long bg = TBG(r);
                    bool ok = false;
                    int pos = 0x7fffffff;
                    if ((bg >> lane) & 1ull) {
                        const int grank = nG - (TCG(r) + __popcll(bg & ltm));
                        pos = (int)(PL ? Pl[i] : Pg[i]);
                        ok = Lk(grank) < pos;
                    }
                    const unsigned long long bo = __ballot(ok);
                    wk += __popcll(bo);
                    if (bo) wmin = min(wmin, __shfl(pos, __ffsll((long long)bo) - 1));
                }
                if (lane == 0) {
                    if (wk) lds_add((lds_i32*)&cs->K, wk);
                    if (wmin != 0x7fffffff) lds_min((lds_i32*)&cs->minR, wmin);
                }
            }
            __syncthreads();
            const int K = cs->K;
            // T[k] <- the entry at L_k, for the entries with key <= kp among the first K L places
            for (int r = w; r < R; r += SW) {
                const int i = a1 + r * 64 + lane;
                const uint32_t v = i < b ? V[i] : 0u;
                if (v != 0u && key_of(v) <= kp) {
                    const unsigned long long ba = TBA(r);
                    const int lrank = (int)(PL ? Pl[i] : Pg[i]) - lo - (TCA(r) + __popcll(ba & ltm)) + 1;
                    if (lrank <= K) W.T[lrank] = (uint32_t)i;
                }
            }
            if (t == ST - 1) cs->cut = K == 0 ? Lk(1) : min(Lk(K + 1), cs->minR);
            __syncthreads();
            const int cut = cs->cut;
            const int off = cs->e_used;
            // output: the left part [f, cut) dense into E, the right part's new list (ordered: entry order = position order)
            //   a swapped R entry (rank k <= K) goes to L_k; its place takes the entry T[k] (or becomes a zero)
            //   a swapped L entry is written by the R entry it changes places with
            //   the others stay: left of the cut into E, right of it into the new list
            // pass A: E and the emit ballots (reusing the "above" tables), pass B: the new list
            int wseeds = 0;
            for (int r = w; r < R; r += SW) {
                const int i = a1 + r * 64 + lane;
                const uint32_t v = i < b ? V[i] : 0u;
                bool emit = false;
                bool left = false;
                if (v != 0u) {
                    const uint32_t key = key_of(v);
                    const unsigned long long ba = TBA(r), bg = TBG(r);
                    const int pos = (int)(PL ? Pl[i] : Pg[i]);
                    const int grank = key >= kp ? nG - (TCG(r) + __popcll(bg & ltm)) : 0x7fffffff;
                    const int lrank = key <= kp ? pos - lo - (TCA(r) + __popcll(ba & ltm)) + 1 : 0x7fffffff;
                    if (grank <= K) { W.E[off + Lk(grank) - f] = v; left = true; emit = W.T[grank] != 0xffffffffu; }
                    else if (lrank <= K) { }
                    else if (pos < cut) { W.E[off + pos - f] = v; left = true; }
                    else emit = true;
                }
                wseeds += __popcll(__ballot(left && (v & 0xfffffu) != 0u));
                const unsigned long long be = __ballot(emit);
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) { if (PL) { lBA[r] = be; lCA[r] = __popcll(be); } else { gBA[r] = be; gCA[r] = __popcll(be); } }
            }
            if (lane == 0 && wseeds) lds_add((lds_i32*)&cs->lseeds, wseeds);
            if (t == 0) W.E[off] = pv;
            __syncthreads();
            if (w == 0) { const int tot = PL ? wave_scan_rows(lCA, R, lane) : wave_scan_rows(gCA, R, lane); if (lane == 0) cs->nright = tot; }
            __syncthreads();
            uint32_t* Pn = PL ? W.P[0] : Pg2;                                // the new list's positions (staged in global memory for the LDS form)
            for (int r = w; r < R; r += SW) {
                const int i = a1 + r * 64 + lane;
                const unsigned long long be = TBA(r);
                if ((be >> lane) & 1ull) {
                    const uint32_t v = V[i];
                    const unsigned long long bg = TBG(r);
                    const int grank = key_of(v) >= kp ? nG - (TCG(r) + __popcll(bg & ltm)) : 0x7fffffff;
                    const int dst = a1 + TCA(r) + __popcll(be & ltm);
                    Pn[dst] = PL ? Pl[i] : Pg[i];
                    V2[dst] = grank <= K ? V[W.T[grank]] : v;
                }
            }
            __syncthreads();
            const int nright = cs->nright;
            if (PL) {
                for (int i = a1 + t; i < a1 + nright; i += ST) Pl[i] = W.P[0][i];
            }
            if (t == 0) {
                const int k = cs->n_ranges;
                cs->ranges[k].off = off; cs->ranges[k].len = cut - f; cs->ranges[k].depth = depth - 1;
                cs->n_ranges = k + 1;
                cs->e_used = off + (cut - f);
                cs->seeds = seeds - cs->lseeds;
                cs->f = cut; cs->depth = depth - 1; cs->a = a1; cs->b = a1 + nright; cs->cur = cur ^ 1;
            }
            __syncthreads();
#undef TBA
#undef TBG
#undef TCA
#undef TCG
#ifdef LF_SEED_STAMPS
            if (t == 0) atomicAdd(&g_dbg_chain[blockIdx.x % 8][1], 1);
#endif
        }
    }
    __syncthreads();
}

// one stable 4-bit counting pass (same scheme as k_lsd_order.hip's radix_pass: every thread owns a contiguous run; [16][ST] counters)
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

// The whole sort of one problem, given its explicit list (m entries; positions in W.P[0] -- copied into LDS when they fit --,
// values in W.V[0]) over an array of n elements with n_seeds seeds: chain, dense phases, and the final insertion sort as stable
// counting passes.  Leaves the seeds in W.out as (n_bins - 1 - bin) << 20 | payload - 1, in their final order.
__device__ __forceinline__ void seed32_sort(const SeedWork& W, int n, int m, int n_seeds, int n_bins, uint32_t* lds, int rows_cap, int* tot, int* base)
{
    __shared__ ChainState cstate;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    SEED_T(t0);
    if (m <= kListLds) {
        lds_u32* Pl = as_lds<lds_u32>(lds);
        for (int i = t; i < m; i += ST) Pl[i] = W.P[0][i];
        __syncthreads();
        sparse_chain<true>(W, Pl, Pl + kListLds, &cstate, n, m, n_seeds);
    } else {
        sparse_chain<false>(W, nullptr, nullptr, &cstate, n, m, n_seeds);
    }
    SEED_T(t1);
    const int M = cstate.e_used;
    introsort_loop_wg(W.E, M, cstate.ranges, cstate.n_ranges, W.dscratch, lds, rows_cap, W.PP);
    SEED_T(t2);
    // ---- the seeds in array order ...
    uint32_t* A = W.out;
    uint32_t* B = W.PP;
    int* rowc = reinterpret_cast<int*>(W.T);                         // [rows + 1] (global: M / 64 rows)
    const int R = (M + 63) >> 6;
    for (int r = w; r < R; r += SW) {
        const int i = r * 64 + lane;
        const uint32_t v = i < M ? W.E[i] : 0u;
        const unsigned long long bb = __ballot((v & 0xfffffu) != 0u);
        if (lane == 0) rowc[r] = __popcll(bb);
    }
    __syncthreads();
    if (w == 0) (void)wave_scan_rows(rowc, R, lane);
    __syncthreads();
    for (int r = w; r < R; r += SW) {
        const int i = r * 64 + lane;
        const uint32_t v = i < M ? W.E[i] : 0u;
        const bool seed = (v & 0xfffffu) != 0u;
        const unsigned long long bb = __ballot(seed);
        if (seed) B[rowc[r] + __popcll(bb & ((1ull << lane) - 1ull))] = ((uint32_t)((n_bins - 1) - (int)key_of(v)) << 20) | ((v & 0xfffffu) - 1u);
    }
    __syncthreads();
    // ... and the final insertion sort: stable by bin, highest bin first (every seed is in E once)
    uint32_t* cnt = lds;
    seed_radix_pass(B, A, n_seeds, 20, cnt, tot, base);
    seed_radix_pass(A, B, n_seeds, 24, cnt, tot, base);
    seed_radix_pass(B, A, n_seeds, 28, cnt, tot, base);
#ifdef LF_SEED_STAMPS
    { const long long t3 = (long long)wall_clock64(); if (t == 0 && blockIdx.x < 6) printf("[seed32] problem %d: n %d list %d seeds %d | chain %lld (%d folds, %d splits -> %d dense in %d ranges)  loop %lld (%d big partitions, %d small ranges; global %lld  blocks %lld  small %lld)  final %lld (x10 ns)\n", (int)blockIdx.x, n, m, n_seeds, t1 - t0, g_dbg_chain[blockIdx.x % 8][0], g_dbg_chain[blockIdx.x % 8][1], M, cstate.n_ranges, t2 - t1, g_dbg_big[blockIdx.x % 8], g_dbg_small[blockIdx.x % 8], g_dbg_t[blockIdx.x % 8][0], g_dbg_t[blockIdx.x % 8][1], g_dbg_t[blockIdx.x % 8][2], t3 - t2); }
#endif
}

// LDS plane of the list's positions: the explicit list in position order is the RANK of every entry (lsd_bitplane.h's layout)
__device__ __forceinline__ void plane_scan(uint32_t* lds, int words, int* wave_tot)
{
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long* bits64 = reinterpret_cast<unsigned long long*>(lds);
    uint16_t* pref = reinterpret_cast<uint16_t*>(lds + 2 * words);
    const int pairs = words >> 1;
    const int per = (pairs + ST - 1) / ST;
    const int w0 = tid * per < pairs ? tid * per : pairs, w1 = w0 + per < pairs ? w0 + per : pairs;
    int mine = 0;
    for (int q = w0; q < w1; ++q) mine += __builtin_popcountll(bits64[2 * q]) + __builtin_popcountll(bits64[2 * q + 1]);
    const int incl = wave_incl_scan_i(mine, lane);
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int basev = incl - mine;
    for (int k = 0; k < wave; ++k) basev += wave_tot[k];
    for (int q = w0; q < w1; ++q) { pref[q] = (uint16_t)basev; basev += __builtin_popcountll(bits64[2 * q]) + __builtin_popcountll(bits64[2 * q + 1]); }
    __syncthreads();
}

__device__ __forceinline__ SeedWork seed_work(uint32_t* sa, uint32_t* sb, uint32_t* oa, uint32_t* ob, uint32_t* tt, size_t cap)
{
    SeedWork W;
    W.E = sa; W.V[0] = sa + cap;
    W.V[1] = sb; W.P[0] = sb + cap;
    W.P[1] = oa; W.PP = ob;
    W.T = tt; W.RT = tt + ((cap + 1) & ~(size_t)1);                   // (8-byte aligned: the row tables hold 64-bit ballots)
    W.dscratch = reinterpret_cast<unsigned long long*>(sb);
    W.out = oa;
    return W;
}

// plane_ok: the gradient image's bit plane fits the kernel's LDS
__global__ __launch_bounds__(ST) void k_lsd_seed32(LsdParams p, const int* __restrict__ n_rec, const unsigned long long* __restrict__ maxgrad,
                                                   const uint32_t* __restrict__ c_xy, const double* __restrict__ c_mod,
                                                   const uint32_t* __restrict__ l_addr, double* l_mod, const int* __restrict__ n_low,
                                                   unsigned long long* __restrict__ sort_a, unsigned long long* __restrict__ sort_b,
                                                   uint32_t* __restrict__ order_a, uint32_t* __restrict__ order_b, int rows_cap, int plane_ok)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t seed_lds[];
    __shared__ int tot[SNB];
    __shared__ int base[SNB];
    __shared__ int wave_tot[SW];
    __shared__ int n_list;
    const int pc = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const size_t Ps = (size_t)p.Hs * p.Ws;
    const size_t o = (size_t)pc * Ps;
    const int nd = n_rec[pc];
    if (nd == 0) return;
    const SeedWork W = seed_work(reinterpret_cast<uint32_t*>(sort_a + o), reinterpret_cast<uint32_t*>(sort_b + o), order_a + o, order_b + o,
                                 reinterpret_cast<uint32_t*>(l_mod + o), Ps);
    const int Wg = p.Ws - 1, Hg = p.Hs - 1;
    const int n = Wg * Hg;
    const double max_grad = __longlong_as_double((long long)maxgrad[pc]);
    const double bin_coef = (max_grad > 0) ? (double)(p.n_bins - 1) / max_grad : 0;
    const int nl = n_low[pc];
    // ---- the explicit list in position order: defined pixels (already in raster order) + the undefined ones with a non-zero bin
    if (plane_ok && nd + nl <= 65535) {
        const int words = bitplane_words((size_t)n);
        for (int i = t; i < 2 * words; i += ST) seed_lds[i] = 0u;
        if (t == 0) n_list = 0;
        __syncthreads();
        for (int e = t; e < nd; e += ST) {
            const uint32_t xy = c_xy[o + e];
            const int pos = (int)(xy >> 16) * Wg + (int)(xy & 0xffffu);
            atomicOr(seed_lds + (pos >> 5), 1u << (pos & 31));
        }
        for (int j = t; j < nl; j += ST) {
            const uint32_t a = l_addr[o + j];
            if ((int)(l_mod[o + j] * bin_coef) > 0) {
                const int y = (int)(a / (uint32_t)p.Ws), x = (int)(a - (uint32_t)y * (uint32_t)p.Ws);
                const int pos = y * Wg + x;
                atomicOr(seed_lds + (pos >> 5), 1u << (pos & 31));
            }
        }
        __syncthreads();
        plane_scan(seed_lds, words, wave_tot);
        int mine = 0;
        for (int e = t; e < nd; e += ST) {
            const uint32_t xy = c_xy[o + e];
            const int pos = (int)(xy >> 16) * Wg + (int)(xy & 0xffffu);
            const uint32_t idx = bitplane_rank(seed_lds, (size_t)n, pos);
            W.P[0][idx] = (uint32_t)pos;
            W.V[0][idx] = ((uint32_t)(int)(c_mod[o + e] * bin_coef) << 20) | (uint32_t)(e + 1);
        }
        for (int j = t; j < nl; j += ST) {
            const uint32_t a = l_addr[o + j];
            const int bin = (int)(l_mod[o + j] * bin_coef);
            if (bin > 0) {
                const int y = (int)(a / (uint32_t)p.Ws), x = (int)(a - (uint32_t)y * (uint32_t)p.Ws);
                const int pos = y * Wg + x;
                const uint32_t idx = bitplane_rank(seed_lds, (size_t)n, pos);
                W.P[0][idx] = (uint32_t)pos;
                W.V[0][idx] = (uint32_t)bin << 20;
                ++mine;
            }
        }
        for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
        if (lane == 0 && mine) atomicAdd(&n_list, mine);
        __syncthreads();
    } else {
        // no plane: the low records (position << 10 | bin) sorted by position with counting passes in global memory, then merged
        // with the defined pixels by rank (binary searches)
        uint32_t* LA = W.V[1];
        uint32_t* LB = W.P[1];
        if (t == 0) n_list = 0;
        __syncthreads();
        // compaction of the records with a non-zero bin (order irrelevant: they are sorted next)
        for (int j0 = 0; j0 < nl; j0 += ST) {
            const int j = j0 + t;
            uint32_t item = 0;
            bool on = false;
            if (j < nl) {
                const uint32_t a = l_addr[o + j];
                const int bin = (int)(l_mod[o + j] * bin_coef);
                const int y = (int)(a / (uint32_t)p.Ws), x = (int)(a - (uint32_t)y * (uint32_t)p.Ws);
                on = bin > 0;
                item = ((uint32_t)(y * Wg + x) << 10) | (uint32_t)bin;
            }
            const unsigned long long bo = __ballot(on);
            int wbase = 0;
            if (lane == 0 && bo) wbase = atomicAdd(&n_list, __popcll(bo));
            wbase = __shfl(wbase, 0);
            if (on) LA[wbase + __popcll(bo & ((1ull << lane) - 1ull))] = item;
        }
        __syncthreads();
        const int nlz = n_list;
        __syncthreads();
        uint32_t* src = LA;
        uint32_t* dst = LB;
        for (int shift = 10; shift < 32; shift += 4) {
            seed_radix_pass(src, dst, nlz, shift, seed_lds, tot, base);
            uint32_t* x = src; src = dst; dst = x;
        }
        // src: sorted by position.  Ranks: a defined pixel e goes to e + (low records in front of it), a low record j to j + (defined in front)
        for (int e = t; e < nd; e += ST) {
            const uint32_t xy = c_xy[o + e];
            const int pos = (int)(xy >> 16) * Wg + (int)(xy & 0xffffu);
            int x = 0, y = nlz;
            while (x < y) { const int mid = (x + y) >> 1; if ((int)(src[mid] >> 10) < pos) x = mid + 1; else y = mid; }
            W.P[0][e + x] = (uint32_t)pos;
            W.V[0][e + x] = ((uint32_t)(int)(c_mod[o + e] * bin_coef) << 20) | (uint32_t)(e + 1);
        }
        for (int j = t; j < nlz; j += ST) {
            const uint32_t it = src[j];
            const int pos = (int)(it >> 10);
            int x = 0, y = nd;
            while (x < y) { const int mid = (x + y) >> 1; const uint32_t xy = c_xy[o + mid]; if ((int)(xy >> 16) * Wg + (int)(xy & 0xffffu) < pos) x = mid + 1; else y = mid; }
            W.P[0][j + x] = (uint32_t)pos;
            W.V[0][j + x] = (it & 1023u) << 20;
        }
        __syncthreads();
    }
    const int m = nd + n_list;
    __syncthreads();
    seed32_sort(W, n, m, nd, p.n_bins, seed_lds, rows_cap, tot, base);
}

// debug / test entry: std::sort(compare_norm) of n keys: E[i] = key << 20 | i + 1.  Elements with key 0 are the detector's flat
// pixels: anonymous.  Leaves the elements with a non-zero key in out[0 .. count) in the order std::sort leaves them, as
// (1023 - key) << 20 | i, and the count in *count.  work: 12 * cap words, cap = max(n, 1024) rounded up to 64.
__global__ __launch_bounds__(ST) void k_std_sort_debug(const uint32_t* __restrict__ E, uint32_t* __restrict__ work, int n, int cap, int rows_cap, int* __restrict__ count)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t seed_lds[];
    __shared__ int tot[SNB];
    __shared__ int base[SNB];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const SeedWork W = seed_work(work, work + 2 * (size_t)cap, work + 4 * (size_t)cap, work + 5 * (size_t)cap, work + 6 * (size_t)cap, (size_t)cap);
    // the list: compaction of the non-zero keys (already in position order)
    int* rowc = reinterpret_cast<int*>(W.RT);
    const int R = (n + 63) >> 6;
    for (int r = w; r < R; r += SW) {
        const int i = r * 64 + lane;
        const uint32_t v = i < n ? E[i] : 0u;
        const unsigned long long bb = __ballot(key_of(v) != 0u);
        if (lane == 0) rowc[r] = __popcll(bb);
    }
    __syncthreads();
    __shared__ int m_sh;
    if (w == 0) { const int total = wave_scan_rows(rowc, R, lane); if (lane == 0) m_sh = total; }
    __syncthreads();
    for (int r = w; r < R; r += SW) {
        const int i = r * 64 + lane;
        const uint32_t v = i < n ? E[i] : 0u;
        const bool on = key_of(v) != 0u;
        const unsigned long long bb = __ballot(on);
        if (on) { const int idx = rowc[r] + __popcll(bb & ((1ull << lane) - 1ull)); W.P[0][idx] = (uint32_t)i; W.V[0][idx] = v; }
    }
    __syncthreads();
    const int m = m_sh;
    if (t == 0) *count = m;
    if (m == 0) return;
    seed32_sort(W, n, m, m, 1024, seed_lds, rows_cap, tot, base);
}

static size_t seed_lds_bytes(int rows_cap, size_t plane_words)
{
    size_t words = (size_t)rows_cap * 6 + 128;
    if (words < (size_t)SW2 * kWaveWords) words = (size_t)SW2 * kWaveWords;
    if (words < (size_t)kBlkWords) words = (size_t)kBlkWords;
    if (words < (size_t)SNB * ST) words = (size_t)SNB * ST;
    if (words < (size_t)kListLds + (size_t)(kListLds / 64) * 6 + 16) words = (size_t)kListLds + (size_t)(kListLds / 64) * 6 + 16;
    if (words < plane_words) words = plane_words;
    return words * sizeof(uint32_t);
}

// the positions of the list are 21-bit, the payloads 20-bit (compact index + 1)
bool lsd_seed32_supported(const LsdParams& p)
{
    const long long n = (long long)(p.Hs - 1) * (p.Ws - 1);
    // (a pixel with a defined gradient must not fall into bin 0, the anonymous one: its norm exceeds rho, and no 8-bit image has a
    // gradient norm above sqrt(2) * 255 = 360.63)
    return p.n_bins <= 1024 && (long long)p.Hs * p.Ws < (1 << 21) && n >= 1 && (double)(p.n_bins - 1) * p.rho / 360.7 >= 1.0;
}

// big != 0: dense problems are expected (LSD of a gray image: most pixels have a gradient) -- the row tables of the dense phase
// cover the whole image in LDS when they fit
void launch_lsd_seed32(const LsdParams& p, int n_frames, const int* n_rec, const unsigned long long* maxgrad, const uint32_t* c_xy,
                       const double* c_mod, const uint32_t* l_addr, double* l_mod, const int* n_low,
                       unsigned long long* sort_a, unsigned long long* sort_b, uint32_t* order_a, uint32_t* order_b, int big, hipStream_t s)
{
    const long long n = (long long)(p.Hs - 1) * (p.Ws - 1);
    int rows_cap = kRowsLds;
    if (big) {
        const long long full = ((n + 63) / 64 + 1 + 63) / 64 * 64;
        if ((size_t)full * 24 + 512 <= (size_t)kMaxLdsBytes) rows_cap = (int)full;
    }
    const size_t plane = bitplane_lds_words((size_t)n);
    const int plane_ok = plane * 4 <= 24 * 1024;
    const size_t lds = seed_lds_bytes(rows_cap, plane_ok ? plane : 0);
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_lsd_seed32), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_lsd_seed32, dim3(n_frames * 3), dim3(ST), lds, s, p, n_rec, maxgrad, c_xy, c_mod, l_addr, l_mod, n_low,
                       sort_a, sort_b, order_a, order_b, rows_cap, plane_ok);
}

// words of device scratch k_std_sort_debug needs for n elements
size_t std_sort_debug_words(int n) { const size_t cap = ((size_t)(n < 1024 ? 1024 : n) + 63) / 64 * 64; return 12 * cap; }

void launch_std_sort_debug(const uint32_t* E, uint32_t* work, int n, int* count, hipStream_t s)
{
    const size_t cap = ((size_t)(n < 1024 ? 1024 : n) + 63) / 64 * 64;
    const int rows_cap = kRowsLds;
    const size_t lds = seed_lds_bytes(rows_cap, 0);
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_std_sort_debug), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_std_sort_debug, dim3(1), dim3(ST), lds, s, E, work, n, (int)cap, rows_cap, count);
}

}  // namespace lf
