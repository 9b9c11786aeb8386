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
// TWO KERNELS per batch since the second half of round 5 (one workgroup per problem in each): k_lsd_seed32 builds the list and runs the chain
// -- the planes need 42 KB of LDS at 511 x 255 and the batched selects 168 registers --, k_lsd_seed32_dense works off the dense ranges and
// runs the final passes with 28 KB and 94 registers; the dense array and the chain's result (elements used, ranges) wait in global memory
// in between.  As one kernel every problem held the larger footprint of both phases for the whole 0.45 ms, and in a pipeline whose SIMDs are
// full of region-growing waves a stage costs its registers and LDS times its duration (123 k -> 129 k frames/s).
// The dense phase exists in two forms: the wave form (default) and the level-synchronous block engine (-DLF_SEED_ENGINE=1: faster alone,
// slower in the pipeline; tools/seed_engine_ab.sh).
//
// One workgroup per problem (frame, colour).  Elements are u32: bin << 20 | payload (compact index + 1 of a pixel with a
// defined gradient, 0 for the others: only seeds need to be told apart).  A RANGE WITHOUT A SEED IS NEVER PARTITIONED: the loop
// only permutes a range within itself, so what it does to a range that holds no seed cannot be seen in the result.
// The heap sort of a range that used up the depth limit (never seen on image data; tested with adversarial keys through
// lf_debug_std_sort) is libstdc++'s __heap_select + __sort_heap replayed by one lane.
#include <cstdlib>
#include <cstring>
#include "common.h"
#include "lsd_bitplane.h"

namespace lf {

// Diagnostic build only (-DLF_SEED_STAMPS): per-phase cycle counts of the first problems, printed from the kernel
#ifdef LF_SEED_STAMPS
#define SEED_T(v) const long long v = (long long)wall_clock64()
#else
#define SEED_T(v) do { } while (0)
#define FST(k) do { } while (0)
#define FST0 do { } while (0)
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
#ifndef LF_SEED_OCC
#define LF_SEED_OCC 3              // workgroups per CU the register allocation must allow (168 VGPRs: three; 128: four, with spills)
#endif
constexpr int ST = LF_SEED_THREADS;  // threads of the list + chain kernel
constexpr int SW = ST / 64;
// the dense kernel in two forms (A/B): 1 = the block engine (all ranges of a block level by level, 512 threads), 0 = the wave form
// (workgroup partitions down to 1024 elements, then single waves; 256 threads)
#ifndef LF_SEED_ENGINE
#define LF_SEED_ENGINE 0
#endif
#ifndef LF_SEED_DENSE_THREADS
#define LF_SEED_DENSE_THREADS (LF_SEED_ENGINE ? 512 : 256)      // (wave form with two waves per problem: - 1.5 % in the pipeline; with one: does not run)
#endif
#ifndef LF_SEED_DENSE_OCC
#define LF_SEED_DENSE_OCC (LF_SEED_ENGINE ? 2 : 4)      // waves per SIMD the register allocation must allow
#endif
constexpr int DT = LF_SEED_DENSE_THREADS;   // threads of the dense kernel (the block engine: one element per thread and row group)
constexpr int kBlock = LF_SEED_BLOCK; // ranges up to this size are worked off in an LDS block
constexpr int SW2 = LF_SEED_WAVES2;  // waves that work in phase 2 of the wave form (each with a private LDS range)
constexpr int kSmall = 1024;         // wave form: ranges up to this size are one wave's work, in LDS (16 rows)
// wave form: one wave's private LDS in phase 2, 32-bit words: the range, the two place lists (u16), row tables, accumulators, range stack
constexpr int kWaveWords = kSmall + kSmall / 2 + 2 * 18 * 2 + 18 + 20 + 8 + 128;
// wave form, phase 1b (a block of <= kBlock elements in LDS): [block][aliased: the workgroup's place lists (u16) | the waves' private lists, tables,
// stacks][the workgroup's row tables][the block's list of small ranges]
constexpr int kBlkWaveWords = kSmall / 2 + 2 * 18 * 2 + 18 + 20 + 8 + 128;
constexpr int kBlkX = SW2 * kBlkWaveWords > kBlock / 2 ? SW2 * kBlkWaveWords : kBlock / 2;
constexpr int kBlkRows = kBlock / 64 + 2;
[[maybe_unused]] constexpr int kBlkWords = kBlock + kBlkX + (kBlkRows * 6 + 8) + 2 * (kBlock / 16);

constexpr int kSortThreshold = 16;   // libstdc++ _S_threshold
constexpr int kMaxLdsBytes = 150 * 1024;
constexpr int SNB = 16;              // buckets of the final counting passes
constexpr int kU = 8;                // rows / pairs in flight per wave / lane in the streaming passes
constexpr int kEPT = LF_SEED_EPT;    // list entries per thread of the LDS form of the chain
constexpr int kListLds = kEPT * ST;  // the explicit list's positions stay in LDS up to this many entries
constexpr int kChainDense = 1024;    // a sparse range this short is written out densely
constexpr int kMaxRanges = 48;       // dense ranges the chain can leave (one per split + the last: the depth allowance is 2 lg n <= 42)
constexpr int kRowsLds = 512;        // row tables in LDS for dense ranges of up to 64 x this many elements (longer ones: tables in global memory)

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

// Wave scans and reductions as DPP (row shifts inside the rows of 16 lanes, then row broadcasts): ~50 cycles instead of six trips through
// the LDS crossbar (__shfl_up compiles to ds_bpermute_b32: ~100 cycles apiece in a lone wave, and the partitions are chains of them)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_or_zero(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false); }
__device__ __forceinline__ int wave_incl_scan_i(int v, int /*lane*/)
{
    v += dpp_or_zero<0x111, 0xf>(v);          // row_shr:1 (lanes without a source add 0)
    v += dpp_or_zero<0x112, 0xf>(v);          // row_shr:2
    v += dpp_or_zero<0x114, 0xf>(v);          // row_shr:4
    v += dpp_or_zero<0x118, 0xf>(v);          // row_shr:8
    v += dpp_or_zero<0x142, 0xa>(v);          // row_bcast:15 into rows 1 and 3
    v += dpp_or_zero<0x143, 0xc>(v);          // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ int wave_last(int v) { return __builtin_amdgcn_readlane(v, 63); }
__device__ __forceinline__ int wave_sum_i(int v) { return wave_last(wave_incl_scan_i(v, 0)); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_or_max(int v) { return __builtin_amdgcn_update_dpp(0x7fffffff, v, CTRL, ROW_MASK, 0xf, false); }
__device__ __forceinline__ int wave_min_i(int v)
{
    v = min(v, dpp_or_max<0x111, 0xf>(v));
    v = min(v, dpp_or_max<0x112, 0xf>(v));
    v = min(v, dpp_or_max<0x114, 0xf>(v));
    v = min(v, dpp_or_max<0x118, 0xf>(v));
    v = min(v, dpp_or_max<0x142, 0xa>(v));
    v = min(v, dpp_or_max<0x143, 0xc>(v));
    return wave_last(v);
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
            carry += wave_last(inc);
        }
    }
    if (w == (COOP ? 1 : 0)) {
        int carry = 0;
        for (int r0 = 0; r0 < R; r0 += 64) {
            const int r = R - 1 - (r0 + lane);                 // from the last row backwards
            const int c = r >= 0 ? SX[r] : 0;
            const int inc = wave_incl_scan_i(c, lane);
            if (r >= 0) SX[r] = carry + inc;
            carry += wave_last(inc);
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
template <int NT, typename T64, typename T32>
__device__ __forceinline__ int partition_global(uint32_t* E, int f, int l, T64 BL, T64 BR, T32 PL, T32 SX,
                                                lds_i32* acc, uint32_t* Lpos, uint32_t* Rpos, int w, int lane)
{
    const int lo = f + 1, hi = l;
    const int R = (hi - lo + 63) >> 6;
    const uint32_t pivot = E[f];
    if (threadIdx.x == 0) { acc[0] = 0; acc[1] = 0x7fffffff; acc[2] = 0x7fffffff; acc[3] = (pivot & 0xfffffu) != 0u; }
    __syncthreads();
    // (1) eight rows in flight per wave
    for (int r0 = w * kU; r0 < R; r0 += (NT / 64) * kU) {
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
    return partition_tail<true>(E, 0, lo, hi, BL, BR, PL, SX, acc, Lpos, Rpos, w, NT / 64, lane, (int)threadIdx.x, NT);
}

#ifdef LF_SEED_STAMPS
__device__ int g_dbg_big[8], g_dbg_small[8], g_dbg_chain[8][4];
__device__ long long g_dbg_b[8][6];
__device__ long long g_dbg_w[8][8];
__device__ long long g_dbg_f[8][12];
#define FST(k) do { const long long n_ = (long long)wall_clock64(); if (threadIdx.x == 0) g_dbg_f[blockIdx.x % 8][k] += n_ - fs_; fs_ = n_; } while (0)
#define FST0 long long fs_ = (long long)wall_clock64()
__device__ long long g_dbg_t[8][4];
#endif

// a dense range the chain left: elements E[off, off + len), depth allowance left
struct SeedRange { int off, len, depth; };

// ---- the wave form of the dense phase (rounds 4 - 5) ------------------------------------------------------------------------------
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
    for (int r = w; r < R; r += SW2) {
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
    return partition_tail<true>(D, 0, lo, hi, BL, BR, PL, SX, acc, Lpos, Rpos, w, SW2, lane, (int)threadIdx.x, 64 * SW2);
}


// A range of at most 64 elements and its whole subtree in ONE wave's registers (round 5): lane = element, every lane carries the
// bounds [f, l) of the range of the loop it currently belongs to, and all ranges inside the window are partitioned AT THE SAME TIME,
// level by level: the median of three by lane shuffles, the L / R ballots masked with the lane's own range, ranks = popcounts, the
// swapped elements change places through 128 words of private LDS under their ranks (range start + rank: the ranges are
// disjoint), the cut from the ballots.  No row tables, no place lists, no stack: ~110 vector instructions per LEVEL whatever the
// number of ranges -- the ~65 % of all partitions that work on 17 .. 64 elements cost ~1.2 k cycles each as single LDS partitions.
__device__ __forceinline__ void wave_window(lds_u32* D, uint32_t* D_generic, int w0, int n, int depth, lds_u32* xch, int lane)
{
    const bool in = lane < n;
    uint32_t v = in ? D[w0 + lane] : 0u;
    int f = 0, l = n, d = depth;
    bool done = false;
    const unsigned long long le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);      // lanes <= this one
    const unsigned long long lt = (1ull << lane) - 1ull;                                // lanes < this one
    for (;;) {
        const unsigned long long segm = (l >= 64 ? ~0ull : ((1ull << l) - 1ull)) & ~((1ull << f) - 1ull);
        const unsigned long long sb = __ballot(in && (v & 0xfffffu) != 0u);
        const bool active = in && !done && (l - f) > kSortThreshold && (sb & segm) != 0ull;
        if (__ballot(active) == 0ull) break;
        if (__ballot(active && d == 0) != 0ull) {                          // depth limit used up: libstdc++ heap sorts the range
            if (in) D[w0 + lane] = v;
            team_sync<false>();
            if (active && d == 0 && lane == f) heap_sort_range(D_generic, w0 + f, w0 + l);
            team_sync<false>();
            if (in) v = D[w0 + lane];
            if (active && d == 0) done = true;
            continue;
        }
        // __move_median_to_first(f, f + 1, mid, l - 1)
        const int ia = f + 1, ib = f + ((l - f) >> 1), ic = l - 1;
        const uint32_t va = __shfl(v, ia), vb = __shfl(v, ib), vc = __shfl(v, ic), vf = __shfl(v, f);
        int pick;
        if (comp(va, vb)) pick = comp(vb, vc) ? ib : (comp(va, vc) ? ic : ia);
        else pick = comp(va, vc) ? ia : (comp(vb, vc) ? ic : ib);
        const uint32_t pv = pick == ia ? va : (pick == ib ? vb : vc);
        if (active) { if (lane == f) v = pv; else if (lane == pick) v = vf; }
        // __unguarded_partition of [f + 1, l) around pv
        const bool part = active && lane > f;
        const uint32_t kp = key_of(pv), kv = key_of(v);
        const bool isL = part && kv <= kp, isR = part && kv >= kp;
        const unsigned long long bl = __ballot(isL) & segm, br = __ballot(isR) & segm;
        const int kl = __popcll(bl & le), r_right = __popcll(br & ~le);
        const int kr = __popcll(br >> lane), l_left = __popcll(bl & lt);
        const bool swl = isL && r_right >= kl, swr = isR && l_left >= kr;
        if (swl) xch[f + kl] = v;
        if (swr) xch[64 + f + kr] = v;
        team_sync<false>();
        if (swl) v = xch[64 + f + kl];
        if (swr) v = xch[f + kr];
        team_sync<false>();
        const unsigned long long bs = __ballot(swl) & segm, bsr = __ballot(swr) & segm;
        const int K = __popcll(bs);
        const unsigned long long un = bl & ~bs;
        const int firstL = bl ? __ffsll((long long)bl) - 1 : 64, firstStay = un ? __ffsll((long long)un) - 1 : 64;
        const int RK = bsr ? __ffsll((long long)bsr) - 1 : 64;
        const int cut = K == 0 ? firstL : min(firstStay, RK);
        if (active) { if (lane < cut) l = cut; else f = cut; --d; }
    }
    if (in) D[w0 + lane] = v;
    team_sync<false>();
}

// __move_median_to_first + __unguarded_partition of the range [f, l) (65 .. 1024 elements: up to 16 rows of 64) of an LDS array by ONE
// wave, without tables in LDS (round 5): every lane reads the three candidates itself (same addresses: a broadcast), the L / R
// ballots of row r stay in the registers of LANE r, their prefix / suffix counts are lane scans, a row's ballots and counts come
// back as v_readlane with the (uniform) row number, the cut is followed in scalar registers while the rows go by (first L, first L
// that stays, leftmost swapped R), four rows' loads are in flight at a time.  What is left in LDS: the two place lists of the
// swapped elements (u16, under their ranks) and the pairwise swaps.  Returns the cut, -1 when the range holds no seed.
__device__ __forceinline__ unsigned long long readlane64(unsigned long long x, int r)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, r), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), r);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ int wave_partition(lds_u32* D, int f, int l, lds_u16* Lpos, lds_u16* Rpos, int lane)
{
    const int lo = f + 1, hi = l;
    const int R = (hi - lo + 63) >> 6;
    const int ia = f + 1, ib = f + ((l - f) >> 1), ic = l - 1;
    const uint32_t ea = D[ia], eb = D[ib], ec = D[ic], ef = D[f];
    int pick;
    if (comp(ea, eb)) pick = comp(eb, ec) ? ib : (comp(ea, ec) ? ic : ia);
    else pick = comp(ea, ec) ? ia : (comp(eb, ec) ? ic : ib);
    const uint32_t pivot = pick == ia ? ea : (pick == ib ? eb : ec);
    if (lane == 0) { D[pick] = ef; D[f] = pivot; }
    team_sync<false>();
    const uint32_t kp = key_of(pivot);
    const unsigned long long le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);      // lanes <= this one
    const unsigned long long lt = (1ull << lane) - 1ull;                                // lanes < this one
    // (1) the ballots of every row; lane r keeps row r's
    unsigned long long myBL = 0ull, myBR = 0ull;
    unsigned long long seeds = (pivot & 0xfffffu) != 0u ? 1ull : 0ull;
    for (int r0 = 0; r0 < R; r0 += 4) {
        uint32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = lo + (r0 + u) * 64 + lane; v[u] = i < hi ? D[i] : 0xffffffffu; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + u;
            const bool valid = lo + r * 64 + lane < hi;
            const unsigned long long bl = __ballot(valid && key_of(v[u]) <= kp);
            const unsigned long long br = __ballot(valid && key_of(v[u]) >= kp);
            seeds |= __ballot(valid && (v[u] & 0xfffffu) != 0u);
            if (lane == r) { myBL = bl; myBR = br; }
        }
    }
    if (seeds == 0ull) return -1;
    // (2) L elements in the rows in front of row r, R elements in the rows behind it
    const int cl = __popcll(myBL), cr = __popcll(myBR);
    const int myPL = wave_incl_scan_i(cl, lane) - cl;
    const int incr = wave_incl_scan_i(cr, lane);
    const int mySX = wave_last(incr) - incr;
    // (3) ranks; the swapped elements publish their places under their ranks; the cut's ingredients in passing
    int K = 0, firstL = 0x7fffffff, firstStay = 0x7fffffff, RK = 0x7fffffff;
    for (int r = 0; r < R; ++r) {
        const unsigned long long bl = readlane64(myBL, r), br = readlane64(myBR, r);
        const int pl = __builtin_amdgcn_readlane(myPL, r), sx = __builtin_amdgcn_readlane(mySX, r);
        const int i = lo + r * 64 + lane;
        const bool isL = (bl >> lane) & 1ull, isR = (br >> lane) & 1ull;
        const int kl = pl + __popcll(bl & le), r_right = sx + __popcll(br & ~le);
        const int kr = sx + __popcll(br >> lane), l_left = pl + __popcll(bl & lt);
        const bool swl = isL && r_right >= kl, swr = isR && l_left >= kr;
        if (swl) Lpos[kl - 1] = (uint16_t)i;
        if (swr) Rpos[kr - 1] = (uint16_t)i;
        const unsigned long long bs = __ballot(swl), bsr = __ballot(swr);
        const unsigned long long un = bl & ~bs;
        K += __popcll(bs);
        if (firstL == 0x7fffffff && bl) firstL = lo + r * 64 + __ffsll((long long)bl) - 1;
        if (firstStay == 0x7fffffff && un) firstStay = lo + r * 64 + __ffsll((long long)un) - 1;
        if (RK == 0x7fffffff && bsr) RK = lo + r * 64 + __ffsll((long long)bsr) - 1;
    }
    team_sync<false>();
    // (4) the swaps, pair by pair
    for (int k0 = lane; k0 < K; k0 += 64 * 4) {
        int pi[4], pj[4];
        uint32_t a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int k = k0 + u * 64; pi[u] = k < K ? (int)Lpos[k] : -1; pj[u] = k < K ? (int)Rpos[k] : -1; }
#pragma unroll
        for (int u = 0; u < 4; ++u) if (pi[u] >= 0) { a[u] = D[pi[u]]; b[u] = D[pj[u]]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) if (pi[u] >= 0) { D[pi[u]] = b[u]; D[pj[u]] = a[u]; }
    }
    team_sync<false>();
    return K == 0 ? firstL : min(firstStay, RK);
}

// one wave works off the whole subtree of the range [f, l) of an LDS array alone (private tables, lists and stack)
__device__ __forceinline__ void wave_subtree(lds_u32* D, uint32_t* D_generic, int f, int l, int depth, lds_u64* wBL, lds_u64* wBR,
                                             lds_i32* wPL, lds_i32* wSX, lds_i32* wacc, lds_u16* Lp, lds_u16* Rp, lds_i32* stack, int lane)
{
    int sp = 0;
    for (;;) {
        while (l - f > kSortThreshold) {
#ifdef LF_SEED_STAMPS
            if (l - f <= 64) { const long long c0 = clock64(); wave_window(D, D_generic, f, l - f, depth, (lds_u32*)Lp, lane); if (lane == 0 && (threadIdx.x >> 6) == 0) { g_dbg_w[blockIdx.x % 8][0] += 1; g_dbg_w[blockIdx.x % 8][1] += clock64() - c0; } break; }
#else
            if (l - f <= 64) { wave_window(D, D_generic, f, l - f, depth, (lds_u32*)Lp, lane); break; }
#endif
            if (depth == 0) {
                if (lane == 0) heap_sort_range(D_generic, f, l);
                team_sync<false>();
                break;
            }
            --depth;
#ifdef LF_SEED_STAMPS
            const long long c0 = clock64();
#endif
            const int cut = wave_partition(D, f, l, Lp, Rp, lane);
#ifdef LF_SEED_STAMPS
            if (lane == 0 && (threadIdx.x >> 6) == 0) { g_dbg_w[blockIdx.x % 8][2] += 1; g_dbg_w[blockIdx.x % 8][3] += clock64() - c0; g_dbg_w[blockIdx.x % 8][4] += (l - f + 62) >> 6; }
#endif
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



// The introsort loop over the listed ranges of E (n = their total extent): the phases 1, 1b, 2 of the dense form.  scratch: global,
// 3 n / 4 + 192 u64 entries (the list of small ranges, the list of blocks, then the two place lists of the global partitions);
// gtab: global row tables (6 words per 64 elements of the longest range + 16) for ranges beyond rows_cap rows.
template <int NT>
__device__ __forceinline__ void introsort_loop_waves(uint32_t* E, int n, const SeedRange* ranges, int n_ranges, unsigned long long* scratch,
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
            if (((l - f - 1 + 63) >> 6) <= rows_cap - 1) cut = partition_global<NT>(E, f, l, BL, BR, PL, SX, acc, Lpos, Rpos, w, lane);
            else {
                const int rc = ((l - f) >> 6) + 2;
                unsigned long long* gBL = reinterpret_cast<unsigned long long*>(gtab);
                unsigned long long* gBR = gBL + rc;
                int* gPL = reinterpret_cast<int*>(gBR + rc);
                int* gSX = gPL + rc;
                cut = partition_global<NT>(E, f, l, gBL, gBR, gPL, gSX, acc, Lpos, Rpos, w, lane);
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
            SEED_T(b0);
            if (t == 0) { acc[3] = 0; n_big = 0; n_blk_small = 0; next_small = 0; }
            __syncthreads();
            {
                int seeds = 0;
                for (int x0 = t; x0 < m; x0 += NT * kU) {
                    uint32_t v[kU];
#pragma unroll
                    for (int u = 0; u < kU; ++u) { const int x = x0 + u * NT; v[u] = x < m ? E[gf + x] : 0u; }
#pragma unroll
                    for (int u = 0; u < kU; ++u) { const int x = x0 + u * NT; if (x < m) { Dg[x] = v[u]; seeds |= (v[u] & 0xfffffu) != 0u; } }
                }
                if (__ballot(seeds) && lane == 0) lds_or(&acc[3], 1);
            }
            __syncthreads();
            if (acc[3] == 0) continue;                                  // no seed in the block
            SEED_T(b1);
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
            SEED_T(b2);
#ifdef LF_SEED_STAMPS
            long long my_w = 0;
#endif
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
                    idx = __builtin_amdgcn_readfirstlane(idx);
                    if (idx >= total) break;
                    const uint32_t a = blk_small[2 * idx], b = blk_small[2 * idx + 1];
                    wave_subtree(Dg, lds, (int)a, (int)(b & 0xffffffu), (int)(b >> 24), wBL, wBR, wPL, wSX, wacc, Lp, Rp, stack, lane);
                }
#ifdef LF_SEED_STAMPS
                my_w = (long long)wall_clock64() - b2;
#endif
            }
            __syncthreads();
            SEED_T(b3);
#ifdef LF_SEED_STAMPS
            if (t == 0) { long long* g = g_dbg_b[blockIdx.x % 8]; g[0] += b1 - b0; g[1] += b2 - b1; g[2] += b3 - b2; g[4] += n_blk_small; }
            if (lane == 0) atomicMin((unsigned long long*)&g_dbg_b[blockIdx.x % 8][5], (unsigned long long)my_w);
#endif
            for (int x0 = t; x0 < m; x0 += NT * kU) {
#pragma unroll
                for (int u = 0; u < kU; ++u) { const int x = x0 + u * NT; if (x < m) E[gf + x] = Dg[x]; }
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
            idx = __builtin_amdgcn_readfirstlane(idx);
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
// The block engine (round 5, second form of the dense phase).  Ranges of up to kBlock elements are gathered into an LDS block --
// several at a time -- and ALL the ranges of the loop inside the block are partitioned AT THE SAME TIME, level by level, by the whole
// workgroup: element x of the block is thread ((x >> 6) % waves, x & 63)'s, which carries the bounds [f, l) and the depth allowance
// of the range x currently belongs to in one register (the loop's ranges are disjoint: no table of ranges exists).  One level:
//   (a) the leader (x == f) of every live range reads the three candidates and swaps the median to the front (__move_median_to_first);
//   (b) every element compares itself with its range's pivot D[f]; the L / R ballots of its row become two bit planes, kept as 32-bit
//       words beside their running popcount: the number of L (R) elements in front of any position is ONE 8-byte read, a mask and a
//       count;
//   (d) ranks from the planes: L elements in front of x minus those in front of f + 1, R elements from x to l; an L element of rank k
//       is swapped iff k R elements lie to its right, an R element of rank k iff k L elements lie to its left (the header's rule); both
//       publish their place under their rank in ONE u16 list (L under f + k, R under l - k: 2 K < l - f), and the first candidate of
//       every (row, range) lowers the range's cut (min of the first L that stays and the leftmost swapped R);
//   (e) the swapped elements fetch their partner's value, (f) store it, read the cut and shrink their bounds to [f, cut) or [cut, l).
// Five barriers per level whatever the number of ranges, ~log2(n / 16) + a few levels; rows without a live range are skipped.  A range
// whose depth allowance is used up is heap sorted by its leader lane (libstdc++'s fallback; adversarial inputs only).
// It replaces the workgroup partitions in an LDS block, the single-wave partitions and the 64-element register windows (which cost
// 5.3 k and 3.6 k cycles apiece as chains of dependent LDS round trips in a lone wave: 160 us of a 4 k-element problem's 200).
#ifdef LF_SEED_STAMPS
#define ENG_T0 long long et_ = (long long)wall_clock64(); if (threadIdx.x == 0) g_dbg_w[blockIdx.x % 8][0] += 1
#define ENG_T(k) do { const long long n_ = (long long)wall_clock64(); if (threadIdx.x == 0) g_dbg_w[blockIdx.x % 8][k] += n_ - et_; et_ = n_; } while (0)
#else
#define ENG_T0 do { } while (0)
#define ENG_T(k) do { } while (0)
#endif
constexpr int kEngRows = kBlock / 64;
constexpr int kEngWordsP = 2 * kEngRows;                 // 32-bit words of a plane
constexpr int kEngCut = kBlock / 16;                     // one cut slot per 16 positions: ranges are longer than 16 and disjoint
constexpr int kEngBatch = 64;                            // ranges per block at most
[[maybe_unused]] constexpr int kEngWords = kBlock + kBlock / 2 + 4 * (kEngWordsP + 2) + 2 * kEngCut + 8 + (kEngBatch + 2) + 2 * kEngBatch;
static_assert(kBlock <= 8192 && kBlock % 512 == 0, "13-bit positions, whole rows per wave");
typedef uint32_t eng_u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) eng_u32x2 lds_u32x2;

__device__ __forceinline__ uint32_t eng_pack(int f, int l, int d) { return (uint32_t)f | ((uint32_t)(l - 1) << 13) | ((uint32_t)d << 26); }
// elements of a plane in front of position q, given the plane's entry (bits, count in front of the word) for q >> 5
__device__ __forceinline__ int eng_rank(eng_u32x2 e, int q) { return (int)e.y + __popc(e.x & ((1u << (q & 31)) - 1u)); }

template <int NT>
__device__ __forceinline__ void dense_blocks(uint32_t* E, const unsigned long long* list, int n_items, uint32_t* lds)
{
    constexpr int NW = NT / 64, RPT = kEngRows / NW, G = RPT < 4 ? RPT : 4;
    static_assert(kEngRows % NW == 0 && RPT % G == 0 && RPT <= 32, "rows per thread");
    lds_u32* D = as_lds<lds_u32>(lds);
    lds_u16* PP = (lds_u16*)(D + kBlock);
    lds_u32x2* LT = (lds_u32x2*)(D + kBlock + kBlock / 2);
    lds_u32x2* RT = LT + (kEngWordsP + 2);
    lds_i32* cutT = (lds_i32*)(RT + (kEngWordsP + 2));
    lds_i32* flags = cutT + 2 * kEngCut;                  // [0], [1] per level: 1 = a live range, 2 = one without depth allowance; [2] seeds; [3] ranges; [4] elements
    lds_i32* boff = flags + 8;                            // [ranges + 1] block offsets
    lds_i32* bsrc = boff + (kEngBatch + 2);               // where a range lies in E
    lds_i32* bdep = bsrc + kEngBatch;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (int i = t; i < kEngWordsP + 2; i += NT) { const eng_u32x2 z = { 0u, 0u }; LT[i] = z; RT[i] = z; }
    int next = 0;
    while (next < n_items) {
        __syncthreads();
        // ---- the next block: as many listed ranges as fit (wave 0)
        if (w == 0) {
            const int i = next + lane;
            const unsigned long long it = i < n_items ? list[i] : 0ull;
            const int rf = (int)(it & 0xffffffu), rl = (int)((it >> 24) & 0xffffffu);
            const int sz = i < n_items ? rl - rf : 0;
            const int inc = wave_incl_scan_i(sz, lane);
            const bool fits = i < n_items && inc <= kBlock;
            const int cnt = __popcll(__ballot(fits));             // sizes are positive: the fitting ones are a prefix
            if (fits) { boff[lane] = inc - sz; bsrc[lane] = rf; bdep[lane] = (int)(it >> 48); }
            if (lane == cnt - 1) { boff[cnt] = inc; flags[4] = inc; }
            if (lane == 0) { flags[3] = cnt; flags[2] = 0; flags[0] = 0; flags[1] = 0; }
        }
        for (int i = t; i < kEngCut; i += NT) cutT[i] = 0x7fffffff;
        __syncthreads();
        const int cnt = flags[3], total = flags[4];
        next += cnt;
        // ---- copy in; every element's range
        uint32_t st[RPT], tmp[RPT];
        int gidx[RPT];
        {
            int seeds = 0, fl = 0;
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                const int x = ((j * NW + w) << 6) | lane;
                int k = 0;
                if (cnt > 1) {
                    int a = 0, b = cnt;                                   // the last range with boff <= x
                    while (b - a > 1) { const int mid = (a + b) >> 1; if (boff[mid] <= x) a = mid; else b = mid; }
                    k = a;
                }
                const bool in = x < total;
                const int f = boff[k], l = boff[k + 1], d = bdep[k];
                gidx[j] = in ? bsrc[k] + (x - f) : -1;
                st[j] = in ? eng_pack(f, l, d) : 0u;
                if (in) fl |= d == 0 ? 3 : 1;
            }
#pragma unroll
            for (int j = 0; j < RPT; ++j) tmp[j] = gidx[j] >= 0 ? E[gidx[j]] : 0u;
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                const int x = ((j * NW + w) << 6) | lane;
                if (gidx[j] >= 0) { D[x] = tmp[j]; seeds |= (tmp[j] & 0xfffffu) != 0u; }
            }
            const unsigned long long sb = __ballot(seeds != 0);
            fl = (__ballot(fl & 1) ? 1 : 0) | (__ballot(fl & 2) ? 2 : 0);
            if (lane == 0) { if (sb) lds_or(&flags[2], 1); if (fl) lds_or(&flags[0], fl); }
        }
        __syncthreads();
        if (flags[2] == 0) continue;                                      // no seed in the block: nothing to order
        // ---- the levels
        for (int lvl = 0;; ++lvl) {
            ENG_T0;
            const int p = lvl & 1;
            lds_i32* cut_cur = cutT + p * kEngCut;
            lds_i32* cut_nxt = cutT + (p ^ 1) * kEngCut;
            const int fl = flags[p];
            if (!(fl & 1)) break;
            for (int i = t; i < kEngCut; i += NT) cut_nxt[i] = 0x7fffffff;
            if (t == 0) flags[p ^ 1] = 0;
            if (fl & 2) {                                                 // depth limit used up somewhere: libstdc++ heap sorts that range
#pragma unroll
                for (int j = 0; j < RPT; ++j) {
                    const uint32_t s = st[j];
                    const int x = ((j * NW + w) << 6) | lane;
                    const int f = (int)(s & 0x1fffu), l = (int)((s >> 13) & 0x1fffu) + 1;
                    if (l - f > kSortThreshold && (s >> 26) == 0u && x == f) heap_sort_range(lds, f, l);
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < RPT; ++j) {
                    const uint32_t s = st[j];
                    const int f = (int)(s & 0x1fffu), l = (int)((s >> 13) & 0x1fffu) + 1;
                    if (l - f > kSortThreshold && (s >> 26) == 0u) st[j] = 0u;
                }
            }
            // (a) the leaders: the median of three to the front
#pragma unroll
            for (int g = 0; g < RPT; g += G) {
                bool lead[G];
                int f[G], l[G];
                bool any = false;
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const uint32_t s = st[g + u];
                    const int x = (((g + u) * NW + w) << 6) | lane;
                    f[u] = (int)(s & 0x1fffu); l[u] = (int)((s >> 13) & 0x1fffu) + 1;
                    lead[u] = l[u] - f[u] > kSortThreshold && x == f[u];
                    any |= lead[u];
                }
                if (__ballot(any) == 0ull) continue;
                uint32_t ea[G], eb[G], ec[G], ef[G];
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const int x = (((g + u) * NW + w) << 6) | lane;
                    ea[u] = D[lead[u] ? f[u] + 1 : x]; eb[u] = D[lead[u] ? f[u] + ((l[u] - f[u]) >> 1) : x]; ec[u] = D[lead[u] ? l[u] - 1 : x]; ef[u] = D[x];
                }
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const int ia = f[u] + 1, ib = f[u] + ((l[u] - f[u]) >> 1), ic = l[u] - 1;
                    int pick;
                    if (comp(ea[u], eb[u])) pick = comp(eb[u], ec[u]) ? ib : (comp(ea[u], ec[u]) ? ic : ia);
                    else pick = comp(ea[u], ec[u]) ? ia : (comp(eb[u], ec[u]) ? ic : ib);
                    const uint32_t pv = pick == ia ? ea[u] : (pick == ib ? eb[u] : ec[u]);
                    if (lead[u]) { D[f[u]] = pv; D[pick] = ef[u]; }
                }
            }
            __syncthreads();
            ENG_T(1);
            // (b) L / R planes
            uint32_t lmask = 0u, rmask = 0u;
#pragma unroll
            for (int g = 0; g < RPT; g += G) {
                bool act[G];
                int f[G];
                bool any = false;
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const uint32_t s = st[g + u];
                    f[u] = (int)(s & 0x1fffu);
                    act[u] = (int)((s >> 13) & 0x1fffu) + 1 - f[u] > kSortThreshold;
                    any |= act[u];
                }
                if (__ballot(any) == 0ull) continue;
                uint32_t v[G], pv[G];
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const int x = (((g + u) * NW + w) << 6) | lane;
                    v[u] = D[x]; pv[u] = D[act[u] ? f[u] : x];
                }
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const int row = (g + u) * NW + w;
                    const int x = (row << 6) | lane;
                    const bool part = act[u] && x > f[u];
                    const uint32_t kp = key_of(pv[u]), kv = key_of(v[u]);
                    const bool isL = part && kv <= kp, isR = part && kv >= kp;
                    const unsigned long long bl = __ballot(isL), br = __ballot(isR);
                    if (isL) lmask |= 1u << (g + u);
                    if (isR) rmask |= 1u << (g + u);
                    if (lane == 0) { LT[2 * row].x = (uint32_t)bl; LT[2 * row + 1].x = (uint32_t)(bl >> 32); RT[2 * row].x = (uint32_t)br; RT[2 * row + 1].x = (uint32_t)(br >> 32); }
                }
            }
            __syncthreads();
            ENG_T(2);
            // (d) running counts of the planes (every wave writes the same table), ranks, places, cuts
            {
                int cL = 0, cR = 0;
#pragma unroll
                for (int r0 = 0; r0 < kEngWordsP; r0 += 64) {
                    const int a = __popc(LT[r0 + lane].x), b = __popc(RT[r0 + lane].x);
                    const int ia = wave_incl_scan_i(a, lane), ib = wave_incl_scan_i(b, lane);
                    LT[r0 + lane].y = (uint32_t)(cL + ia - a); RT[r0 + lane].y = (uint32_t)(cR + ib - b);
                    cL += wave_last(ia); cR += wave_last(ib);
                }
                if (lane == 0) { LT[kEngWordsP].y = (uint32_t)cL; RT[kEngWordsP].y = (uint32_t)cR; }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
            }
            uint32_t swm = 0u;
#pragma unroll
            for (int g = 0; g < RPT; g += G) {
                bool act[G];
                int f[G], l[G];
                bool any = false;
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const uint32_t s = st[g + u];
                    f[u] = (int)(s & 0x1fffu); l[u] = (int)((s >> 13) & 0x1fffu) + 1;
                    act[u] = l[u] - f[u] > kSortThreshold;
                    any |= act[u];
                }
                if (__ballot(any) == 0ull) continue;
                eng_u32x2 lx[G], rx[G], lf[G], rl[G];
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const int x = (((g + u) * NW + w) << 6) | lane;
                    lx[u] = LT[x >> 5]; rx[u] = RT[x >> 5];
                    lf[u] = LT[act[u] ? (f[u] + 1) >> 5 : x >> 5]; rl[u] = RT[act[u] ? l[u] >> 5 : x >> 5];
                }
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const int x = (((g + u) * NW + w) << 6) | lane;
                    const bool isL = (lmask >> (g + u)) & 1u, isR = (rmask >> (g + u)) & 1u;
                    const int l_left = eng_rank(lx[u], x) - eng_rank(lf[u], f[u] + 1), kl = l_left + 1;
                    const int kr = eng_rank(rl[u], l[u]) - eng_rank(rx[u], x), r_right = kr - (isR ? 1 : 0);
                    const bool swl = isL && r_right >= kl, swr = isR && l_left >= kr;
                    if (swl) PP[f[u] + kl] = (uint16_t)x;
                    if (swr) PP[l[u] - kr] = (uint16_t)x;
                    tmp[g + u] = swl ? (uint32_t)(l[u] - kl) : (swr ? (uint32_t)(f[u] + kr) : 0u);
                    if (swl || swr) swm |= 1u << (g + u);
                    // the first candidate of this range in this row lowers the cut
                    const bool cand = (isL && !swl) || swr;
                    const unsigned long long cm = __ballot(cand);
                    const int before = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0u));
                    const int at_start = __builtin_amdgcn_ds_bpermute(4 * max(f[u] - (x & ~63), 0), before);
                    if (cand && before == at_start) lds_min(&cut_cur[f[u] >> 4], x);
                }
            }
            __syncthreads();
            ENG_T(3);
            // (e) the partner's value
#pragma unroll
            for (int g = 0; g < RPT; g += G) {
                if (__ballot(((swm >> g) & ((1u << G) - 1u)) != 0u) == 0ull) continue;
                int y[G];
#pragma unroll
                for (int u = 0; u < G; ++u) y[u] = (int)PP[((swm >> (g + u)) & 1u) ? tmp[g + u] : 0u];
#pragma unroll
                for (int u = 0; u < G; ++u) { const uint32_t vy = D[y[u]]; if ((swm >> (g + u)) & 1u) tmp[g + u] = vy; }
            }
            __syncthreads();
            // (f) the swap, the cut, the bounds of the next level
            {
                int more = 0;
#pragma unroll
                for (int g = 0; g < RPT; g += G) {
                    bool act[G];
                    int f[G], l[G], cut[G];
                    bool any = false;
#pragma unroll
                    for (int u = 0; u < G; ++u) {
                        const uint32_t s = st[g + u];
                        f[u] = (int)(s & 0x1fffu); l[u] = (int)((s >> 13) & 0x1fffu) + 1;
                        act[u] = l[u] - f[u] > kSortThreshold;
                        any |= act[u];
                    }
                    if (__ballot(any) == 0ull) continue;
#pragma unroll
                    for (int u = 0; u < G; ++u) cut[u] = cut_cur[act[u] ? f[u] >> 4 : 0];
#pragma unroll
                    for (int u = 0; u < G; ++u) {
                        const int x = (((g + u) * NW + w) << 6) | lane;
                        if ((swm >> (g + u)) & 1u) D[x] = tmp[g + u];
                        if (act[u]) {
                            const int d = (int)(st[g + u] >> 26) - 1;
                            const int nf = x < cut[u] ? f[u] : cut[u], nl = x < cut[u] ? cut[u] : l[u];
                            st[g + u] = eng_pack(nf, nl, d);
                            if (nl - nf > kSortThreshold) more |= d == 0 ? 3 : 1;
                        }
                    }
                }
                more = (__ballot(more & 1) ? 1 : 0) | (__ballot(more & 2) ? 2 : 0);
                if (lane == 0 && more) lds_or(&flags[p ^ 1], more);
            }
            __syncthreads();
            ENG_T(4);
        }
        // ---- copy out
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const int x = ((j * NW + w) << 6) | lane;
            if (gidx[j] >= 0) tmp[j] = D[x];
        }
#pragma unroll
        for (int j = 0; j < RPT; ++j) if (gidx[j] >= 0) E[gidx[j]] = tmp[j];
    }
    __syncthreads();
}


// The introsort loop over the listed ranges of E (n = their total extent).  scratch: global, 3 n / 4 + 192 u64 entries (the list of
// ranges of <= kBlock elements, then the two place lists of the global partitions); gtab: global row tables (6 words per 64 elements
// of the longest range + 16) for ranges beyond rows_cap rows.
template <int NT>
__device__ __forceinline__ void introsort_loop_wg(uint32_t* E, int n, const SeedRange* ranges, int n_ranges, unsigned long long* scratch,
                                                  uint32_t* lds, int rows_cap, uint32_t* gtab)
{
    // LDS: phase 1 = the row tables of the global partitions; phase 2 = the block engine's (aliased)
    lds_u64* BL = as_lds<lds_u64>(lds);
    lds_u64* BR = BL + rows_cap;
    lds_i32* PL = (lds_i32*)(BR + rows_cap);
    lds_i32* SX = PL + rows_cap;                          // rows_cap + 1 entries
    unsigned long long* work_list = scratch;                           // ranges of 17 .. kBlock elements: phase 2
    uint32_t* Lpos = reinterpret_cast<uint32_t*>(scratch + (n / 16 + 64) + (n / 1024 + 64));
    uint32_t* Rpos = Lpos + (n / 2 + 8);
    __shared__ int acc_[4];
    lds_i32* acc = as_lds<lds_i32>(acc_);
    __shared__ int big_stack[3 * (72 + kMaxRanges)];
    __shared__ int n_big, n_work;
    __shared__ int cur[4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    auto pack = [](int f, int l, int depth) { return (unsigned long long)f | ((unsigned long long)l << 24) | ((unsigned long long)depth << 48); };
    if (t == 0) {
        n_big = 0; n_work = 0;
        for (int i = 0; i < n_ranges; ++i) {
            const int f = ranges[i].off, l = f + ranges[i].len, d = ranges[i].depth, sz = ranges[i].len;
            if (sz > kBlock) { big_stack[3 * n_big] = f; big_stack[3 * n_big + 1] = l; big_stack[3 * n_big + 2] = d; ++n_big; }
            else if (sz > kSortThreshold) work_list[n_work++] = pack(f, l, d);
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
            if (((l - f - 1 + 63) >> 6) <= rows_cap - 1) cut = partition_global<NT>(E, f, l, BL, BR, PL, SX, acc, Lpos, Rpos, w, lane);
            else {
                const int rc = ((l - f) >> 6) + 2;
                unsigned long long* gBL = reinterpret_cast<unsigned long long*>(gtab);
                unsigned long long* gBR = gBL + rc;
                int* gPL = reinterpret_cast<int*>(gBR + rc);
                int* gSX = gPL + rc;
                cut = partition_global<NT>(E, f, l, gBL, gBR, gPL, gSX, acc, Lpos, Rpos, w, lane);
            }
#ifdef LF_SEED_STAMPS
            if (t == 0) atomicAdd(&g_dbg_big[blockIdx.x % 8], 1);
#endif
            if (cut < 0) { l = f; break; }                           // no seed in the range: nothing to order
            if (t == 0) {
                const int rs = l - cut;
                if (rs > kBlock) { big_stack[3 * n_big] = cut; big_stack[3 * n_big + 1] = l; big_stack[3 * n_big + 2] = depth; ++n_big; }
                else if (rs > kSortThreshold) work_list[n_work++] = pack(cut, l, depth);
            }
            l = cut;
            __syncthreads();
        }
        if (t == 0 && l - f > kSortThreshold) work_list[n_work++] = pack(f, l, depth);
        __syncthreads();
    }
    SEED_T(tb);
    // ---- phase 2: everything else, block by block in LDS
    const int nw = n_work;
    __syncthreads();
    dense_blocks<NT>(E, work_list, nw, lds);
#ifdef LF_SEED_STAMPS
    { const long long tc = (long long)wall_clock64(); if (t == 0) { g_dbg_t[blockIdx.x % 8][0] = tb - ta; g_dbg_t[blockIdx.x % 8][1] = tc - tb; g_dbg_small[blockIdx.x % 8] = nw; } }
#endif
}

// ============================================================================================================================
// The chain: the top of the introsort loop on the explicit list (header).  One problem's global work areas, each of at least
// `cap` u32 words (cap >= the array length n and >= 1024):
struct SeedWork {
    uint32_t* E;              // the dense array the chain writes out
    uint32_t* V0; uint32_t* V1;   // the list's values (bin << 20 | seed + 1), double buffered (scalar members, selected with ?: -- an array indexed
    uint32_t* P0; uint32_t* P1;   // by the buffer number put the struct into scratch memory and made every access a flat one); the list's positions
    uint32_t* PP;             // split: the positions of the entries above the pivot; the dense phase's global row tables; the final passes' second buffer
    uint32_t* T;              // split: the entry at the k-th L place (or none)
    uint32_t* RT;             // row tables of a list that does not fit LDS (6 words per 64 entries)
    unsigned long long* dscratch;   // the dense phase's range lists and place lists: aliases V1 | P0
    uint32_t* out;            // the seeds in their final order (aliases P1)
    int cap;                  // entries each of these areas holds (LsdParams::rec_cap)
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
        carry += wave_last(inc);
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

// ---- the chain on a BIT PLANE (ranges whose plane fits LDS; round 5, second form) -------------------------------------------------
// Nothing in a fold or a split needs the list in ORDER -- only "how many entries lie in front of position q" (a rank) and "where
// is the k-th position without an entry" (a select) -- so for a range of up to a few hundred thousand positions the list stays an
// unordered array of (position, value) in global memory and a step works on the bit plane of the entries' positions in LDS
// (lsd_bitplane.h's layout: 64-bit words + a 16-bit running count per pair of words; rebuilt per step: the range halves with
// every fold): a fold rewrites nothing but the POSITIONS of the entries it moves (values never move, nothing is merged or
// sorted), a split reads two planes (the entries above the pivot / not below it).  rank = 3 LDS reads, select = a binary search
// over the running counts of zeros + a bisection inside the word.
struct Plane { lds_u64* bits; lds_u16* pref; lds_u16* stab; int pairs; };
// words of LDS one plane over `len` positions takes: the bits, a count per pair of words, a select entry per 128 clear bits
__host__ __device__ inline int plane_words_for(int len) { const int words = ((len >> 6) + 2) & ~1, pairs = words >> 1; return 2 * words + ((pairs + 1) >> 1) + ((pairs + 3) >> 1); }
__device__ __forceinline__ Plane plane_at(lds_u32* base, int len)
{
    Plane pl;
    const int words = ((len >> 6) + 2) & ~1;
    pl.bits = (lds_u64*)base;
    pl.pairs = words >> 1;
    pl.pref = (lds_u16*)(base + 2 * words);
    pl.stab = (lds_u16*)(base + 2 * words + ((pl.pairs + 1) >> 1));
    return pl;
}
// set bits in front of bit q
__device__ __forceinline__ int plane_rank(const Plane& pl, int q)
{
    const int w = q >> 6;
    const unsigned long long lo = pl.bits[w & ~1], cur = pl.bits[w];
    return (int)pl.pref[w >> 1] + ((w & 1) ? __popcll(lo) : 0) + __popcll(cur & ((1ull << (q & 63)) - 1ull));
}
// the kk-th set bit of wd (1-based): the largest p with fewer than kk set bits below bit p
__device__ __forceinline__ int word_select(unsigned long long wd, int kk)
{
    int p = 0;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        if (__popcll(wd & ((1ull << (p + s)) - 1ull)) < kk) p += s;
    }
    return p;
}
// N selects at a time, step by step together (a lone wave waits ~100+ cycles for every LDS read: N chains side by side cost
// little more than one): out[u] = index of the k[u]-th (1-based) CLEAR bit; the caller makes sure the plane holds that many.
// stab[j] = the pair of words that holds clear bit number 128 j + 1, so the pair of the k-th is at most a few steps further.
template <int N>
__device__ __forceinline__ void plane_select0_n(const Plane& pl, const int (&k)[N], int (&out)[N])
{
    int x[N];
#pragma unroll
    for (int u = 0; u < N; ++u) x[u] = (int)pl.stab[(k[u] - 1) >> 7];
    for (;;) {                                                   // on to the last pair with fewer than k clear bits in front of it
        int nx[N];
        bool any = false;
#pragma unroll
        for (int u = 0; u < N; ++u) { const int c = x[u] + 1; nx[u] = c < pl.pairs ? 128 * c - (int)pl.pref[c] : 0x7fffffff; }
#pragma unroll
        for (int u = 0; u < N; ++u) if (nx[u] < k[u]) { ++x[u]; any = true; }
        if (!any) break;
    }
    unsigned long long w0[N], w1[N];
    int p0[N];
#pragma unroll
    for (int u = 0; u < N; ++u) { w0[u] = pl.bits[2 * x[u]]; w1[u] = pl.bits[2 * x[u] + 1]; p0[u] = (int)pl.pref[x[u]]; }
#pragma unroll
    for (int u = 0; u < N; ++u) {
        int kk = k[u] - (128 * x[u] - p0[u]);
        unsigned long long wd = ~w0[u];
        int base = 128 * x[u];
        const int z0 = __popcll(wd);
        if (kk > z0) { kk -= z0; wd = ~w1[u]; base += 64; }
        out[u] = base + word_select(wd, kk);
    }
}
__device__ __forceinline__ int plane_select0(const Plane& pl, int k)
{
    const int kk[1] = { k };
    int out[1];
    plane_select0_n<1>(pl, kk, out);
    return out[0];
}

// running counts of a plane whose bits are set (all threads; ends synchronised); returns the number of set bits
__device__ __forceinline__ int plane_counts(const Plane& pl, lds_i32* wave_tot)
{
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (pl.pairs + ST - 1) / ST;
    const int w0 = tid * per < pl.pairs ? tid * per : pl.pairs, w1 = w0 + per < pl.pairs ? w0 + per : pl.pairs;
    int mine = 0;
    for (int q = w0; q < w1; ++q) mine += __popcll(pl.bits[2 * q]) + __popcll(pl.bits[2 * q + 1]);
    const int incl = wave_incl_scan_i(mine, lane);
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int basev = incl - mine, total = 0;
    for (int k = 0; k < SW; ++k) { const int v = wave_tot[k]; if (k < wave) basev += v; total += v; }
    for (int q = w0; q < w1; ++q) {
        const int c = __popcll(pl.bits[2 * q]) + __popcll(pl.bits[2 * q + 1]);
        pl.pref[q] = (uint16_t)basev;
        // the select table: this pair holds the clear bits number zb0 + 1 .. zb1
        const int zb0 = 128 * q - basev, zb1 = zb0 + 128 - c;
        for (int j = (zb0 + 127) >> 7; 128 * j < zb1; ++j) pl.stab[j] = (uint16_t)q;
        basev += c;
    }
    __syncthreads();
    return total;
}
__device__ __forceinline__ void plane_set(lds_u32* base, int q) { (void)__hip_atomic_fetch_or(base + (q >> 5), 1u << (q & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }


// one step (fold or split) of the chain on the planes.  The list: slots [a, b) of (cur ? W.P1 : W.P0) / (cur ? W.V1 : W.V0), any order, value 0 = dead slot.
// pr / regs_ok: while the list has at most kEPT x ST slots every thread keeps the positions of ITS slots (a + t + u ST) in registers
// from one fold to the next (0xffffffff = dead slot; global memory is written through for the steps that read it): a run of folds --
// the usual top of the chain -- then reads nothing from global memory at all, it is plane arithmetic in LDS.
constexpr uint32_t kDeadSlot = 0xffffffffu;
#ifndef LF_SEED_SEL
#define LF_SEED_SEL 4
#endif
constexpr int kSel = LF_SEED_SEL;           // selects / list slots a thread works on side by side (registers: the kernel must keep three waves per SIMD)
__device__ __forceinline__ void chain_step_plane(const SeedWork& W, __attribute__((address_space(3))) ChainState* cs, lds_u32* lds,
                                                 uint32_t (&pr)[kEPT], bool& regs_ok)
{
    const int t = threadIdx.x, lane = t & 63;
    const int f = cs->f, l = cs->l, a = cs->a, b = cs->b, depth = cs->depth, cur = cs->cur, seeds = cs->seeds;
    const int len = l - f;
    uint32_t* V = (cur ? W.V1 : W.V0);
    uint32_t* P = (cur ? W.P1 : W.P0);
    uint32_t* V2 = (cur ? W.V0 : W.V1);
    uint32_t* P2 = (cur ? W.P0 : W.P1);
    lds_i32* wtot = (lds_i32*)&cs->wtot[0];
    const bool persist = b - a <= kEPT * ST;
    auto load_tile = [&](int base) {
        uint32_t vv[kEPT];
#pragma unroll
        for (int u = 0; u < kEPT; ++u) { const int i = base + t + u * ST, ic = i < b ? i : b - 1; vv[u] = V[ic]; pr[u] = P[ic]; }     // (clamped, not predicated: the loads go out together)
#pragma unroll
        for (int u = 0; u < kEPT; ++u) if (vv[u] == 0u || base + t + u * ST >= b) pr[u] = kDeadSlot;
    };
    // ---- pass A: the plane of the entries' positions over [f, l) (bit = position - f), and the entries at f, f + 1, mid, l - 1
    FST0;
    const Plane pl = plane_at(lds, len);
    for (int i = t; i < 4 * pl.pairs; i += ST) lds[i] = 0u;
    if (t < 4) { cs->md_idx[t] = -1; cs->md_val[t] = 0u; }
    __syncthreads();
    const int q0 = f, q1 = f + 1, q2 = f + len / 2, q3 = l - 1;
    for (int base = a; base < b; base += kEPT * ST) {
        if (!(persist && regs_ok)) load_tile(base);
#pragma unroll
        for (int u = 0; u < kEPT; ++u) {
            const uint32_t pos = pr[u];
            if (pos != kDeadSlot) {
                plane_set(lds, (int)pos - f);
                if ((int)pos == q0 || (int)pos == q1 || (int)pos == q2 || (int)pos == q3) {
                    const int i = base + t + u * ST;
                    const int which = (int)pos == q0 ? 0 : ((int)pos == q1 ? 1 : ((int)pos == q2 ? 2 : 3));
                    cs->md_idx[which] = i; cs->md_val[which] = V[i];
                }
            }
        }
    }
    if (persist) regs_ok = true;
    __syncthreads();
    FST(0);
    const int total = plane_counts(pl, wtot);
    FST(1);
    const uint32_t vf = cs->md_val[0], va = cs->md_val[1], vb = cs->md_val[2], vc = cs->md_val[3];
    const int xf = cs->md_idx[0];
    int pk;                                                            // 1: f + 1, 2: mid, 3: l - 1  (three different places: len > 16)
    if (comp(va, vb)) pk = comp(vb, vc) ? 2 : (comp(va, vc) ? 3 : 1);
    else pk = comp(va, vc) ? 1 : (comp(vb, vc) ? 3 : 2);
    const uint32_t pv = cs->md_val[pk];
    const int xp = cs->md_idx[pk];
    const int pick = pk == 1 ? q1 : (pk == 2 ? q2 : q3);
    __syncthreads();
    // the median goes to f, what was at f to the median's place
    if (t == 0) {
        if (xp >= 0) { V[xp] = vf; if (xf >= 0) V[xf] = 0u; }           // (vf == 0: the slot dies)
        else if (xf >= 0) P[xf] = (uint32_t)pick;
    }
    const int lo = f + 1, hi = l, plen = hi - lo;
    if (key_of(pv) == 0u) {
        // ================================================ FOLD
        if (xf >= 0) {                                                    // the entry at f (bit 0) went to `pick`: its bit moves, the counts are made again (rare)
            if (persist) {
#pragma unroll
                for (int u = 0; u < kEPT; ++u) if (a + t + u * ST == xf) pr[u] = (uint32_t)pick;
            }
            if (t == 0) {
                (void)__hip_atomic_fetch_and(lds, ~1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                plane_set(lds, pick - f);
            }
            __syncthreads();
            (void)plane_counts(pl, wtot);
        }
        __syncthreads();
        // bit 0 (the pivot's place) is clear: the entries of [lo, q) are rank(q - f), the k-th zero of [lo, hi) is the plane's (k + 1)-th
        FST(2);
        const int mm = total;
        const int Zt = plen - mm;
        // K = the largest k with 2 k - (entries inside the last k positions) <= Zt: one wave, 64 candidates per round
        if (t < 64) {
            int klo = 0, khi = plen + 1;                                  // feasible(klo), not feasible(khi)
            while (khi - klo > 1) {
                const int step = (khi - klo + 63) / 64;
                const int k = klo + (lane + 1) * step;
                bool feas = false;
                if (k < khi) feas = 2 * k - (mm - plane_rank(pl, len - k)) <= Zt;
                const unsigned long long bf = __ballot(feas);
                const int cnt = bf == ~0ull ? 64 : __ffsll((long long)~bf) - 1;   // candidates 1 .. cnt are feasible
                const int nlo = klo + cnt * step;
                khi = min(khi, nlo + step);
                klo = nlo;
            }
            if (lane == 0) cs->K = klo;
        }
        __syncthreads();
        FST(3);
        const int K = cs->K;
        if (t == ST - 1) {
            int cut;
            if (K == 0) cut = f + plane_select0(pl, 2);
            else { cut = hi - K; if (Zt >= K + 1) cut = min(cut, f + plane_select0(pl, K + 2)); }
            cs->cut = cut;
        }
        // the entries of the last K positions move to the first K zeros: the one at hi - k to the k-th (eight selects at a time)
        for (int base = a; base < b; base += kEPT * ST) {
            if (!persist) load_tile(base);
#pragma unroll
            for (int g = 0; g < kEPT; g += kSel) {
                int kk[kSel], sel[kSel];
                bool any = false;
#pragma unroll
                for (int u = 0; u < kSel; ++u) {
                    const uint32_t pos = pr[g + u];
                    const bool mv = pos != kDeadSlot && (int)pos >= hi - K;
                    kk[u] = mv ? hi - (int)pos + 1 : 1;
                    any |= mv;
                }
                if (__ballot(any) == 0ull) continue;
                plane_select0_n<kSel>(pl, kk, sel);
#pragma unroll
                for (int u = 0; u < kSel; ++u) {
                    const uint32_t pos = pr[g + u];
                    if (pos != kDeadSlot && (int)pos >= hi - K) {
                        const uint32_t np = (uint32_t)(f + sel[u]);
                        pr[g + u] = np;
                        P[base + t + (g + u) * ST] = np;
                    }
                }
            }
        }
        __syncthreads();
        FST(4);
        if (t == 0) { cs->l = cs->cut; cs->depth = depth - 1; }
        __syncthreads();
        FST(5);
#ifdef LF_SEED_STAMPS
        if (t == 0) atomicAdd(&g_dbg_chain[blockIdx.x % 8][0], 1);
#endif
        return;
    }
    __syncthreads();
    // ================================================ SPLIT
    // Every pass walks the list in tiles of kST x ST slots whose loads go out together (one trip to L2 per tile and pass; a slot at a
    // time was a trip each: 75 us per split).
    const uint32_t kp = key_of(pv);
    const int pw = plane_words_for(plen);
    lds_u32* ldsG = lds + ((pw + 1) & ~1);
    const Plane pa = plane_at(lds, plen), pg = plane_at(ldsG, plen);       // entries above the pivot / not below it
    constexpr int kST = 8;
#pragma unroll
    for (int u = 0; u < kEPT; ++u) pr[u] = kDeadSlot;                     // (the register copy of the positions ends here: a split writes a new list)
    regs_ok = false;
    uint32_t vv[kST], pp[kST];
    auto load_vals = [&](int base) {                                       // (after the median's move: a dead slot has value 0)
#pragma unroll
        for (int u = 0; u < kST; ++u) { const int i = base + t + u * ST, ic = i < b ? i : b - 1; vv[u] = V[ic]; pp[u] = P[ic]; }
#pragma unroll
        for (int u = 0; u < kST; ++u) if (base + t + u * ST >= b) vv[u] = 0u;
    };
    for (int i = t; i < 4 * pa.pairs; i += ST) { lds[i] = 0u; ldsG[i] = 0u; }
    if (t == 0) { cs->K = 0; cs->minR = 0x7fffffff; cs->lseeds = (pv & 0xfffffu) != 0u; cs->nright = 0; }
    __syncthreads();
    for (int base = a; base < b; base += kST * ST) {
        load_vals(base);
#pragma unroll
        for (int u = 0; u < kST; ++u) {
            if (vv[u] && key_of(vv[u]) >= kp) {
                plane_set(ldsG, (int)pp[u] - lo);
                if (key_of(vv[u]) > kp) plane_set(lds, (int)pp[u] - lo);
            }
        }
    }
    __syncthreads();
    FST(6);
    const int nA = plane_counts(pa, wtot);
    const int nG = plane_counts(pg, wtot);
    const int nonA = plen - nA;
    FST(7);
    // K = the R entries (rank k from the right) with L_k in front of them; R_K = the leftmost of them;  TV <- none
    for (int k = t; k <= nG + 1; k += ST) W.T[k] = 0u;
    {
        int wk = 0, mymin = 0x7fffffff;                                   // (one reduction per wave at the end: a shuffle costs a lone wave ~150 cycles)
        for (int base = a; base < b; base += kST * ST) {
            load_vals(base);
#pragma unroll
            for (int g = 0; g < kST; g += kSel) {
                int gr[kSel], kk[kSel], sel[kSel];
                bool any = false;
#pragma unroll
                for (int u = 0; u < kSel; ++u) {
                    const bool isG = vv[g + u] != 0u && key_of(vv[g + u]) >= kp;
                    gr[u] = isG ? nG - plane_rank(pg, (int)pp[g + u] - lo) : 0x7fffffff;
                    kk[u] = gr[u] <= nonA ? gr[u] : 1;
                    any |= isG;
                }
                if (__ballot(any) == 0ull) continue;
                plane_select0_n<kSel>(pa, kk, sel);
#pragma unroll
                for (int u = 0; u < kSel; ++u) {
                    const int pos = (int)pp[g + u];
                    const bool ok = gr[u] <= nonA && lo + sel[u] < pos;
                    wk += __popcll(__ballot(ok));
                    if (ok) mymin = min(mymin, pos);
                }
            }
        }
        mymin = wave_min_i(mymin);
        if (lane == 0 && wk) { lds_add((lds_i32*)&cs->K, wk); lds_min((lds_i32*)&cs->minR, mymin); }
    }
    __syncthreads();
    FST(8);
    const int K = cs->K;
    // TV[k] <- the VALUE of the entry at L_k, for the entries with key <= kp among the first K L places (0: that place holds a zero)
    for (int base = a; base < b; base += kST * ST) {
        load_vals(base);
#pragma unroll
        for (int u = 0; u < kST; ++u) {
            if (vv[u] && key_of(vv[u]) <= kp) {
                const int q = (int)pp[u] - lo;
                const int lrank = q - plane_rank(pa, q) + 1;
                if (lrank <= K) W.T[lrank] = vv[u];
            }
        }
    }
    if (t == ST - 1) {
        const int l1 = K + 1 <= nonA ? lo + plane_select0(pa, K + 1) : 0x7fffffff;
        cs->cut = K == 0 ? l1 : min(l1, cs->minR);
    }
    __syncthreads();
    FST(9);
    const int cut = cs->cut;
    const int off = cs->e_used;
    // output: the left part [f, cut) dense into E, the right part's list (any order) into the other buffers
    //   a swapped R entry (rank k <= K) goes to L_k; its place takes the element that was at L_k (TV[k]; a zero: the slot goes)
    //   a swapped L entry is written by the R entry it changes places with
    //   the others stay: left of the cut into E, right of it into the new list
    int wseeds = 0;
    for (int base = a; base < b; base += kST * ST) {
        load_vals(base);
        int gr[kST];
        uint32_t tv[kST];
        uint32_t lmask = 0u, emask = 0u;                                   // bit u: slot u is a swapped L element / goes into the new list
#pragma unroll
        for (int u = 0; u < kST; ++u) {
            const uint32_t v = vv[u];
            const int q = (int)pp[u] - lo;
            gr[u] = 0x7fffffff;
            if (v) {
                if (key_of(v) >= kp) gr[u] = nG - plane_rank(pg, q);
                if (key_of(v) <= kp && q - plane_rank(pa, q) + 1 <= K) lmask |= 1u << u;
            }
        }
#pragma unroll
        for (int u = 0; u < kST; ++u) tv[u] = gr[u] <= K ? W.T[gr[u]] : 0u;    // (the tile's gathers together)
#pragma unroll
        for (int g = 0; g < kST; g += kSel) {
            int kk[kSel], sel[kSel];
            bool any = false;
#pragma unroll
            for (int u = 0; u < kSel; ++u) { kk[u] = gr[g + u] <= K ? gr[g + u] : 1; any |= gr[g + u] <= K; }
            if (__ballot(any) == 0ull) continue;
            plane_select0_n<kSel>(pa, kk, sel);
#pragma unroll
            for (int u = 0; u < kSel; ++u) if (gr[g + u] <= K) W.E[off + lo + sel[u] - f] = vv[g + u];
        }
        int wemit = 0;
#pragma unroll
        for (int u = 0; u < kST; ++u) {
            const uint32_t v = vv[u];
            const int pos = (int)pp[u];
            bool emit = false, left = false;
            if (v) {
                if (gr[u] <= K) { left = true; emit = tv[u] != 0u; }
                else if ((lmask >> u) & 1u) { }
                else if (pos < cut) { W.E[off + pos - f] = v; left = true; }
                else { emit = true; tv[u] = v; }
            }
            wseeds += __popcll(__ballot(left && (v & 0xfffffu) != 0u));
            wemit += __popcll(__ballot(emit));
            if (emit) emask |= 1u << u;
        }
        if (wemit) {                                                        // one reservation per wave and tile
            int base_ = 0;
            if (lane == 0) base_ = __hip_atomic_fetch_add((lds_i32*)&cs->nright, wemit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            base_ = a + __builtin_amdgcn_readfirstlane(base_);
#pragma unroll
            for (int u = 0; u < kST; ++u) {
                const bool emit = (emask >> u) & 1u;
                const unsigned long long be = __ballot(emit);
                if (emit) { const int dst = base_ + __popcll(be & ((1ull << lane) - 1ull)); P2[dst] = pp[u]; V2[dst] = tv[u]; }
                base_ += __popcll(be);
            }
        }
    }
    regs_ok = false;                                                      // a new list, in the other buffers
    if (lane == 0 && wseeds) lds_add((lds_i32*)&cs->lseeds, wseeds);
    if (t == 0) W.E[off] = pv;
    __syncthreads();
    FST(10);
    if (t == 0) {
        const int k = cs->n_ranges;
        cs->ranges[k].off = off; cs->ranges[k].len = cut - f; cs->ranges[k].depth = depth - 1;
        cs->n_ranges = k + 1;
        cs->e_used = off + (cut - f);
        cs->seeds = seeds - cs->lseeds;
        cs->f = cut; cs->depth = depth - 1; cs->b = a + cs->nright; cs->cur = cur ^ 1;
    }
    __syncthreads();
#ifdef LF_SEED_STAMPS
    if (t == 0) atomicAdd(&g_dbg_chain[blockIdx.x % 8][1], 1);
#endif
}

// PL: the positions in LDS (single buffer Pl, new places through registers), else in (cur ? W.P1 : W.P0) (double buffered)
template <bool PL>
__device__ __forceinline__ void sparse_chain(const SeedWork& W, lds_u32* Pl, lds_u32* tab, ChainState* cs_generic, int n, int m0, int nseeds, lds_u32* plane_lds, int plane_lds_words)
{
    typedef __attribute__((address_space(3))) ChainState lds_cs;
    lds_cs* cs = (lds_cs*)(__attribute__((address_space(3))) void*)cs_generic;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) {
        cs->f = 0; cs->l = n; cs->depth = 2 * (31 - __clz(n)); cs->a = 0; cs->b = m0; cs->seeds = nseeds; cs->cur = 0; cs->e_used = 0; cs->n_ranges = 0; cs->stop = 0;
    }
    uint32_t pr[kEPT];                                                    // the plane form's register copy of this thread's list slots
    bool regs_ok = false;
    __syncthreads();
    for (;;) {
        const int f = cs->f, l = cs->l, a = cs->a, b = cs->b, depth = cs->depth, cur = cs->cur, seeds = cs->seeds;
        const int len = l - f, m = b - a;
        uint32_t* V = (cur ? W.V1 : W.V0);
        uint32_t* V2 = (cur ? W.V0 : W.V1);
        uint32_t* Pg = (cur ? W.P1 : W.P0);
        uint32_t* Pg2 = (cur ? W.P0 : W.P1);
        __syncthreads();                                                  // everybody has read the state
        if (seeds == 0) break;
        if (len <= kChainDense || (len - m) * 4 <= len || depth == 0 || cs->n_ranges >= kMaxRanges - 1) {
            const int off = cs->e_used;
            if (off + len > W.cap) {                                      // the dense array would not hold it (the split parts are explicit entries: they always fit)
                if (t == 0) cs->stop = off + len;
                __syncthreads();
                break;
            }
            if (PL) chain_materialize(W.E, off, f, l, Pl, V, a, b); else chain_materialize(W.E, off, f, l, Pg, V, a, b);
            if (t == 0) { const int k = cs->n_ranges; cs->ranges[k].off = off; cs->ranges[k].len = len; cs->ranges[k].depth = depth; cs->n_ranges = k + 1; cs->e_used = off + len; }
            __syncthreads();
            break;
        }
        if (!PL && 2 * ((plane_words_for(len) + 1) & ~1) <= plane_lds_words && m <= 65535) {   // the range's planes fit LDS: the unordered form
#ifdef LF_SEED_STAMPS
            const long long c0 = (long long)wall_clock64(); const int nr0 = cs->n_ranges;
#endif
            chain_step_plane(W, cs, plane_lds, pr, regs_ok);
#ifdef LF_SEED_STAMPS
            if (t == 0) { const int sp = cs->n_ranges != nr0; g_dbg_chain[blockIdx.x % 8][2 + sp] += (int)((long long)wall_clock64() - c0); }
#endif
            continue;
        }
        // ---- (longer ranges: the list in position order, ranks by binary search) the median of three: the values at f, f + 1, mid, l - 1
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
                    if (bo) wmin = min(wmin, __builtin_amdgcn_readlane(pos, __ffsll((long long)bo) - 1));
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
            uint32_t* Pn = PL ? W.P0 : Pg2;                                // the new list's positions (staged in global memory for the LDS form)
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
                for (int i = a1 + t; i < a1 + nright; i += ST) Pl[i] = W.P0[i];
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
template <int NT>
__device__ __forceinline__ void seed_radix_pass(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int n, int shift,
                                                uint32_t* cnt /*[SNB][NT]*/, int* tot, int* base)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int seg = (n + NT - 1) / NT;
    const int i0 = min(n, t * seg), i1 = min(n, i0 + seg);
    for (int b = 0; b < SNB; ++b) cnt[b * NT + t] = 0;
    for (int i = i0; i < i1; ++i) cnt[(int)((src[i] >> shift) & (SNB - 1)) * NT + t]++;
    __syncthreads();
    for (int bb = 0; bb < SNB / (NT / 64); ++bb) {
        const int b = wave * (SNB / (NT / 64)) + bb;
        int carry = 0;
        for (int c = 0; c < NT / 64; ++c) {
            const int v = (int)cnt[b * NT + c * 64 + lane];
            const int inc = wave_incl_scan_i(v, lane);
            cnt[b * NT + c * 64 + lane] = (uint32_t)(carry + inc - v);
            carry += wave_last(inc);
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
        dst[(uint32_t)base[b] + cnt[b * NT + t]++] = it;
    }
    __syncthreads();
}

// The sort of one problem is two kernels (round 5, second half): k_lsd_seed32 builds the explicit list and runs the CHAIN on it -- bit
// planes of the whole gradient image in LDS, 256 threads --, k_lsd_seed32_dense the introsort loop on the dense ranges the chain left and
// the final insertion sort -- the block engine, 512 threads, half the LDS.  Between them: the dense array E in global memory and the
// chain's result (elements used, ranges) in the first words of the problem's RT area.

// kernel 1: the chain on the explicit list (m entries; positions in W.P0, values in W.V0) over an array of n elements with n_seeds seeds
__device__ __forceinline__ int seed32_chain(const SeedWork& W, int n, int m, int n_seeds, uint32_t* lds, int lds_words)
{
    __shared__ ChainState cstate;
    const int t = threadIdx.x;
    SEED_T(t0);
    sparse_chain<false>(W, nullptr, nullptr, &cstate, n, m, n_seeds, as_lds<lds_u32>(lds), lds_words);
    int* state = reinterpret_cast<int*>(W.RT);
    if (cstate.stop) return cstate.stop;                                  // the dense array needs this many entries
    if (t == 0) { state[0] = cstate.e_used; state[1] = cstate.n_ranges; }
    for (int i = t; i < cstate.n_ranges; i += ST) { state[2 + 3 * i] = cstate.ranges[i].off; state[3 + 3 * i] = cstate.ranges[i].len; state[4 + 3 * i] = cstate.ranges[i].depth; }
#ifdef LF_SEED_STAMPS
    { const long long t1 = (long long)wall_clock64(); if (t == 0 && blockIdx.x < 6) printf("[seed32] problem %d: n %d list %d seeds %d | chain %lld (%d folds %d, %d splits %d -> %d dense in %d ranges) (x10 ns) | fold: passA %lld counts %lld median %lld K %lld passB %lld state %lld | split: build %lld counts %lld K %lld TV %lld out %lld\n", (int)blockIdx.x, n, m, n_seeds, t1 - t0, g_dbg_chain[blockIdx.x % 8][0], g_dbg_chain[blockIdx.x % 8][2], g_dbg_chain[blockIdx.x % 8][1], g_dbg_chain[blockIdx.x % 8][3], cstate.e_used, cstate.n_ranges, g_dbg_f[blockIdx.x % 8][0], g_dbg_f[blockIdx.x % 8][1], g_dbg_f[blockIdx.x % 8][2], g_dbg_f[blockIdx.x % 8][3], g_dbg_f[blockIdx.x % 8][4], g_dbg_f[blockIdx.x % 8][5], g_dbg_f[blockIdx.x % 8][6], g_dbg_f[blockIdx.x % 8][7], g_dbg_f[blockIdx.x % 8][8], g_dbg_f[blockIdx.x % 8][9], g_dbg_f[blockIdx.x % 8][10]); }
#endif
    return 0;
}

// kernel 2: the dense phases and the final insertion sort as stable counting passes.  Leaves the seeds in W.out as
// (n_bins - 1 - bin) << 20 | payload - 1, in their final order.
template <int NT>
__device__ __forceinline__ void seed32_dense(const SeedWork& W, int n_seeds, int n_bins, uint32_t* lds, int rows_cap, int* tot, int* base)
{
    __shared__ SeedRange ranges[kMaxRanges];
    __shared__ int hdr[2];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int* state = reinterpret_cast<const int*>(W.RT);
    if (t < 2) hdr[t] = state[t];
    __syncthreads();
    const int M = hdr[0], n_ranges = hdr[1];
    for (int i = t; i < n_ranges; i += NT) { ranges[i].off = state[2 + 3 * i]; ranges[i].len = state[3 + 3 * i]; ranges[i].depth = state[4 + 3 * i]; }
    __syncthreads();
    SEED_T(t1);
#if LF_SEED_ENGINE
    introsort_loop_wg<NT>(W.E, M, ranges, n_ranges, W.dscratch, lds, rows_cap, W.PP);
#else
    static_assert(NT == 64 * SW2, "the wave form's workgroup");
    introsort_loop_waves<NT>(W.E, M, ranges, n_ranges, W.dscratch, lds, rows_cap, W.PP);
#endif
    SEED_T(t2);
    // ---- the seeds in array order ...
    uint32_t* A = W.out;
    uint32_t* B = W.PP;
    int* rowc = reinterpret_cast<int*>(W.T);                         // [rows + 1] (global: M / 64 rows)
    const int R = (M + 63) >> 6;
    for (int r = w; r < R; r += NT / 64) {
        const int i = r * 64 + lane;
        const uint32_t v = i < M ? W.E[i] : 0u;
        const unsigned long long bb = __ballot((v & 0xfffffu) != 0u);
        if (lane == 0) rowc[r] = __popcll(bb);
    }
    __syncthreads();
    if (w == 0) (void)wave_scan_rows(rowc, R, lane);
    __syncthreads();
    for (int r = w; r < R; r += NT / 64) {
        const int i = r * 64 + lane;
        const uint32_t v = i < M ? W.E[i] : 0u;
        const bool seed = (v & 0xfffffu) != 0u;
        const unsigned long long bb = __ballot(seed);
        if (seed) B[rowc[r] + __popcll(bb & ((1ull << lane) - 1ull))] = ((uint32_t)((n_bins - 1) - (int)key_of(v)) << 20) | ((v & 0xfffffu) - 1u);
    }
    __syncthreads();
    // ... and the final insertion sort: stable by bin, highest bin first (every seed is in E once)
    uint32_t* cnt = lds;
    seed_radix_pass<NT>(B, A, n_seeds, 20, cnt, tot, base);
    seed_radix_pass<NT>(A, B, n_seeds, 24, cnt, tot, base);
    seed_radix_pass<NT>(B, A, n_seeds, 28, cnt, tot, base);
#ifdef LF_SEED_STAMPS
    { const long long t3 = (long long)wall_clock64(); if (t == 0 && blockIdx.x < 6) printf("[seed32 dense] problem %d: %d elements in %d ranges, seeds %d | loop %lld (%d big partitions, %d listed ranges; global %lld  blocks %lld)  final %lld (x10 ns) | %lld levels: a %lld b %lld d %lld e+f %lld\n", (int)blockIdx.x, M, n_ranges, n_seeds, t2 - t1, g_dbg_big[blockIdx.x % 8], g_dbg_small[blockIdx.x % 8], g_dbg_t[blockIdx.x % 8][0], g_dbg_t[blockIdx.x % 8][1], t3 - t2, g_dbg_w[blockIdx.x % 8][0], g_dbg_w[blockIdx.x % 8][1], g_dbg_w[blockIdx.x % 8][2], g_dbg_w[blockIdx.x % 8][3], g_dbg_w[blockIdx.x % 8][4]); }
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
    W.E = sa; W.V0 = sa + cap;
    W.V1 = sb; W.P0 = sb + cap;
    W.P1 = oa; W.PP = ob;
    W.T = tt; W.RT = tt + ((cap + 1) & ~(size_t)1);                   // (8-byte aligned: the row tables hold 64-bit ballots)
    W.dscratch = reinterpret_cast<unsigned long long*>(sb);
    W.out = oa;
    W.cap = (int)cap;
    return W;
}

// plane_ok: the gradient image's bit plane fits the kernel's LDS
__global__ __launch_bounds__(ST, LF_SEED_OCC) void k_lsd_seed32(LsdParams p, int* __restrict__ n_rec, int* __restrict__ norder, int* __restrict__ rec_need,
                                                   const unsigned long long* __restrict__ maxgrad,
                                                   const uint32_t* __restrict__ c_xy, const double* __restrict__ c_mod,
                                                   const uint32_t* __restrict__ l_addr, double* l_mod, const int* __restrict__ n_low,
                                                   unsigned long long* __restrict__ sort_a, unsigned long long* __restrict__ sort_b,
                                                   uint32_t* __restrict__ order_a, uint32_t* __restrict__ order_b, int plane_ok, int lds_words)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t seed_lds[];
    __shared__ int tot[SNB];
    __shared__ int base[SNB];
    __shared__ int wave_tot[SW];
    __shared__ int n_list;
    const int pc = blockIdx.x, t = threadIdx.x, lane = t & 63;
    const size_t Ps = (size_t)p.rec_cap;                                  // entries per problem of every list
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
    // The problem's lists hold rec_cap entries: the explicit list (at most nd + nl) must fit, and so must the low records (k_lsd_grad wrote
    // none of them otherwise).  A problem that does not fit is dropped from this batch (no seeds: k_lsd_grow leaves it alone) and its need
    // reported; the host grows the lists and runs the batch again (lanefront_api.hip: lsd_records_retry).
    if (nl > p.rec_cap || nd + nl > p.rec_cap) {
        if (t == 0) { atomicMax(rec_need, nd + nl); n_rec[pc] = 0; norder[pc] = 0; }
        return;
    }
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
            W.P0[idx] = (uint32_t)pos;
            W.V0[idx] = ((uint32_t)(int)(c_mod[o + e] * bin_coef) << 20) | (uint32_t)(e + 1);
        }
        for (int j = t; j < nl; j += ST) {
            const uint32_t a = l_addr[o + j];
            const int bin = (int)(l_mod[o + j] * bin_coef);
            if (bin > 0) {
                const int y = (int)(a / (uint32_t)p.Ws), x = (int)(a - (uint32_t)y * (uint32_t)p.Ws);
                const int pos = y * Wg + x;
                const uint32_t idx = bitplane_rank(seed_lds, (size_t)n, pos);
                W.P0[idx] = (uint32_t)pos;
                W.V0[idx] = (uint32_t)bin << 20;
                ++mine;
            }
        }
        mine = wave_sum_i(mine);
        if (lane == 0 && mine) atomicAdd(&n_list, mine);
        __syncthreads();
    } else {
        // no plane: the low records (position << 10 | bin) sorted by position with counting passes in global memory, then merged
        // with the defined pixels by rank (binary searches)
        uint32_t* LA = W.V1;
        uint32_t* LB = W.P1;
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
            wbase = __builtin_amdgcn_readfirstlane(wbase);
            if (on) LA[wbase + __popcll(bo & ((1ull << lane) - 1ull))] = item;
        }
        __syncthreads();
        const int nlz = n_list;
        __syncthreads();
        uint32_t* src = LA;
        uint32_t* dst = LB;
        for (int shift = 10; shift < 32; shift += 4) {
            seed_radix_pass<ST>(src, dst, nlz, shift, seed_lds, tot, base);
            uint32_t* x = src; src = dst; dst = x;
        }
        // src: sorted by position.  Ranks: a defined pixel e goes to e + (low records in front of it), a low record j to j + (defined in front)
        for (int e = t; e < nd; e += ST) {
            const uint32_t xy = c_xy[o + e];
            const int pos = (int)(xy >> 16) * Wg + (int)(xy & 0xffffu);
            int x = 0, y = nlz;
            while (x < y) { const int mid = (x + y) >> 1; if ((int)(src[mid] >> 10) < pos) x = mid + 1; else y = mid; }
            W.P0[e + x] = (uint32_t)pos;
            W.V0[e + x] = ((uint32_t)(int)(c_mod[o + e] * bin_coef) << 20) | (uint32_t)(e + 1);
        }
        for (int j = t; j < nlz; j += ST) {
            const uint32_t it = src[j];
            const int pos = (int)(it >> 10);
            int x = 0, y = nd;
            while (x < y) { const int mid = (x + y) >> 1; const uint32_t xy = c_xy[o + mid]; if ((int)(xy >> 16) * Wg + (int)(xy & 0xffffu) < pos) x = mid + 1; else y = mid; }
            W.P0[j + x] = (uint32_t)pos;
            W.V0[j + x] = (it & 1023u) << 20;
        }
        __syncthreads();
    }
    const int m = nd + n_list;
    __syncthreads();
    const int need = seed32_chain(W, n, m, nd, seed_lds, lds_words);
    if (need && t == 0) { atomicMax(rec_need, need); n_rec[pc] = 0; norder[pc] = 0; }
}

// kernel 2 of a problem: see seed32_dense
__global__ __launch_bounds__(DT, LF_SEED_DENSE_OCC) void k_lsd_seed32_dense(LsdParams p, const int* __restrict__ n_rec, double* l_mod,
                                                            unsigned long long* __restrict__ sort_a, unsigned long long* __restrict__ sort_b,
                                                            uint32_t* __restrict__ order_a, uint32_t* __restrict__ order_b, int rows_cap)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t seed_lds[];
    __shared__ int tot[SNB];
    __shared__ int base[SNB];
    const int pc = blockIdx.x;
    const size_t Ps = (size_t)p.rec_cap;
    const size_t o = (size_t)pc * Ps;
    const int nd = n_rec[pc];
    if (nd == 0) return;
    const SeedWork W = seed_work(reinterpret_cast<uint32_t*>(sort_a + o), reinterpret_cast<uint32_t*>(sort_b + o), order_a + o, order_b + o,
                                 reinterpret_cast<uint32_t*>(l_mod + o), Ps);
    seed32_dense<DT>(W, nd, p.n_bins, seed_lds, rows_cap, tot, base);
}

// debug / test entry: std::sort(compare_norm) of n keys: E[i] = key << 20 | i + 1.  Elements with key 0 are the detector's flat
// pixels: anonymous.  Leaves the elements with a non-zero key in out[0 .. count) in the order std::sort leaves them, as
// (1023 - key) << 20 | i, and the count in *count.  work: 12 * cap words, cap = max(n, 1024) rounded up to 64.
__global__ __launch_bounds__(ST, LF_SEED_OCC) void k_std_sort_debug(const uint32_t* __restrict__ E, uint32_t* __restrict__ work, int n, int cap, int* __restrict__ count, int lds_words)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t seed_lds[];
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
        if (on) { const int idx = rowc[r] + __popcll(bb & ((1ull << lane) - 1ull)); W.P0[idx] = (uint32_t)i; W.V0[idx] = v; }
    }
    __syncthreads();
    const int m = m_sh;
    if (t == 0) *count = m;
    if (m == 0) return;
    if (seed32_chain(W, n, m, m, seed_lds, lds_words) && t == 0) *count = 0;      // (cannot happen: cap >= n)
}

__global__ __launch_bounds__(DT, LF_SEED_DENSE_OCC) void k_std_sort_debug_dense(uint32_t* __restrict__ work, int cap, int rows_cap, const int* __restrict__ count)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t seed_lds[];
    __shared__ int tot[SNB];
    __shared__ int base[SNB];
    const int m = *count;
    if (m == 0) return;
    const SeedWork W = seed_work(work, work + 2 * (size_t)cap, work + 4 * (size_t)cap, work + 5 * (size_t)cap, work + 6 * (size_t)cap, (size_t)cap);
    seed32_dense<DT>(W, m, 1024, seed_lds, rows_cap, tot, base);
}

// LDS of the chain kernel: the two planes of a split over the whole array when they fit kPlaneLdsBytes (then the whole chain runs on
// planes; longer arrays start in the ordered form and change over once their range is short enough); at least the counters of the
// counting passes that order a list without a plane
constexpr size_t kPlaneLdsBytes = 56 * 1024;
static size_t seed_lds_bytes(long long n)
{
    size_t words = (size_t)SNB * ST;
    bool planes = false;
    if (n < (1ll << 30)) {
        const size_t two = 2 * (((size_t)plane_words_for((int)n) + 1) & ~(size_t)1);
        planes = two * 4 <= kPlaneLdsBytes;
        if (planes && words < two) words = two;
    }
    if (!planes && words < 8192) words = 8192;                            // room for the planes of a range once it is short enough
    return words * sizeof(uint32_t);
}
// ... of the dense kernel: the block engine, the row tables of the global partitions, the counters of the final passes
static size_t seed_dense_lds_bytes(int rows_cap)
{
    size_t words = (size_t)rows_cap * 6 + 128;
#if LF_SEED_ENGINE
    if (words < (size_t)kEngWords) words = (size_t)kEngWords;
#else
    if (words < (size_t)SW2 * kWaveWords) words = (size_t)SW2 * kWaveWords;
    if (words < (size_t)kBlkWords) words = (size_t)kBlkWords;
#endif
    if (words < (size_t)SNB * DT) words = (size_t)SNB * DT;
    return words * sizeof(uint32_t);
}

// the positions of the list are 21-bit, the payloads 20-bit (compact index + 1)
bool lsd_seed32_supported(const LsdParams& p)
{
    const long long n = (long long)(p.Hs - 1) * (p.Ws - 1);
    // (a pixel with a defined gradient must not fall into bin 0, the anonymous one: its norm exceeds rho, and no 8-bit image has a
    // gradient norm above sqrt(2) * 255 = 360.63)
    // (bins are 12-bit keys above the 20-bit payload: up to 4096 of them)
    return p.n_bins <= 4096 && (long long)p.Hs * p.Ws < (1 << 21) && n >= 1 && (double)(p.n_bins - 1) * p.rho / 360.7 >= 1.0;
}

// big != 0: dense problems are expected (LSD of a gray image: most pixels have a gradient) -- the row tables of the dense phase
// cover the whole image in LDS when they fit
void launch_lsd_seed32(const LsdParams& p, int n_frames, int* n_rec, int* norder, int* rec_need, const unsigned long long* maxgrad, const uint32_t* c_xy,
                       const double* c_mod, const uint32_t* l_addr, double* l_mod, const int* n_low,
                       unsigned long long* sort_a, unsigned long long* sort_b, uint32_t* order_a, uint32_t* order_b, int big, hipStream_t s)
{
    const long long n = (long long)(p.Hs - 1) * (p.Ws - 1);
    int rows_cap = kRowsLds;
    if (big) {
        const long long full = ((n + 63) / 64 + 1 + 63) / 64 * 64;
        if ((size_t)full * 24 + 512 <= (size_t)kMaxLdsBytes) rows_cap = (int)full;
    }
    const size_t lds = seed_lds_bytes(n), lds2 = seed_dense_lds_bytes(rows_cap);
    const int plane_ok = bitplane_lds_words((size_t)n) * 4 <= lds;      // the list's initial order by ranks in a plane of the gradient image
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_lsd_seed32), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (lds2 > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_lsd_seed32_dense), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    static const char* diag_skip = getenv("LF_DIAG_SKIP");     // diagnostic only (what-if timing, results are wrong): "seedchain", "seeddense"
    if (!(diag_skip && strstr(diag_skip, "seedchain")))
    hipLaunchKernelGGL(k_lsd_seed32, dim3(n_frames * 3), dim3(ST), lds, s, p, n_rec, norder, rec_need, maxgrad, c_xy, c_mod, l_addr, l_mod, n_low,
                       sort_a, sort_b, order_a, order_b, plane_ok, (int)(lds / 4));
    if (!(diag_skip && (strstr(diag_skip, "seeddense") || strstr(diag_skip, "seedchain"))))
    hipLaunchKernelGGL(k_lsd_seed32_dense, dim3(n_frames * 3), dim3(DT), lds2, s, p, n_rec, l_mod, sort_a, sort_b, order_a, order_b, rows_cap);
}

// words of device scratch k_std_sort_debug needs for n elements
size_t std_sort_debug_words(int n) { const size_t cap = ((size_t)(n < 1024 ? 1024 : n) + 63) / 64 * 64; return 12 * cap; }

void launch_std_sort_debug(const uint32_t* E, uint32_t* work, int n, int* count, hipStream_t s)
{
    const size_t cap = ((size_t)(n < 1024 ? 1024 : n) + 63) / 64 * 64;
    const int rows_cap = kRowsLds;
    const size_t lds = seed_lds_bytes(n), lds2 = seed_dense_lds_bytes(rows_cap);
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_std_sort_debug), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (lds2 > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_std_sort_debug_dense), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    hipLaunchKernelGGL(k_std_sort_debug, dim3(1), dim3(ST), lds, s, E, work, n, (int)cap, count, (int)(lds / 4));
    hipLaunchKernelGGL(k_std_sort_debug_dense, dim3(1), dim3(DT), lds2, s, work, (int)cap, rows_cap, static_cast<const int*>(count));
}

}  // namespace lf
