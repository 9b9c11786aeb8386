// LSD region growing, rectangle fit, refinement and NFA validation for ONE (frame, colour)
// problem, executed by ONE 64-lane wavefront.
//
// Restates OpenCV 3.x lsd.cpp (flsd main loop, region_grow, region2rect, get_theta, refine,
// reduce_region_radius, rect_improve, rect_nfa, nfa) exactly as the CPU oracle does
// (oracle/lf_oracle_lsd.c) -- reached in the reference through
// /root/reference/src/line_detector/include/line_detector/line_detector_lsd.py:64-72.
//
// Why one wave per problem: region growing is order dependent (seed order, USED flags, the
// running float angle sums), so the sequential order is part of the result.  Parallelism
// comes from (a) 3 x B independent problems per batch, one per SIMD, and (b) inside a
// problem, lanes cooperate wherever the arithmetic is order independent or can be staged:
//   * the 8-neighbour probe loads all neighbours in one memory round trip (lanes 0..8),
//     then the accept/angle-update chain runs on wave-uniform values (v_readlane);
//   * ordered f64 sums (centroid, inertia, refine statistics) prefetch 64 region points per
//     round trip and accumulate in region order from lane registers;
//   * min/max extents, USED-flag resets and the rectangle pixel counts of the NFA test are
//     integer / min-max reductions across lanes (exact, order free).
// Per-pixel data are addressed by COMPACT index (raster rank among defined pixels); LDS holds
// the row starts, the sorted x lists, one USED bit per defined pixel and the head of the region list
// (13 KB for a 512x256 problem), so twelve problems are resident per CU.
//
// The same source compiles for the host (LF_HOST_SIM, one lane) so the control flow can be
// checked against the oracle without a GPU (tests/hostsim); that build is a test harness,
// not a product path.
#pragma once
#include <type_traits>
#include "detmath.h"

#ifndef LF_HOST_SIM
#include <hip/hip_runtime.h>
#define LFG_DEV __device__ __forceinline__
#define LFG_NL 64
#else
#include <math.h>
#include <string.h>
#define LFG_DEV static inline
#define LFG_NL 1
#endif

namespace lf {
namespace grow {

constexpr double PI_ = 3.14159265358979323846;
constexpr double M_3_2_PI_ = (3 * PI_) / 2;
constexpr double M_2__PI_ = 2 * PI_;
constexpr double DEG2RAD = PI_ / 180;
constexpr float NOTDEF_F = -1024.0f;
constexpr double NOTDEF_D = -1024.0;

// ------------------------------------------------------------------ wave primitives
#ifndef LF_HOST_SIM
// the lanes for which p holds, as the condition mask the compiler already has (HIP's __ballot(int) goes through a 0 / 1 register and
// a compare: three instructions and a vector -> scalar dependency per ballot, eight ballots per accept span)
LFG_DEV unsigned long long lfg_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
LFG_DEV int lane_id() { return (int)(threadIdx.x & 63u); }
LFG_DEV int rl_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
LFG_DEV float rl_f(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
LFG_DEV double rl_d(double v, int src)
{
    long long b = __double_as_longlong(v);
    int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src);
    int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// (sums and maxima over the wave use DPP exchanges inside the rows of 16 lanes and scalar registers across the four rows; see
// wave_max_d below)
template <int CTRL>
LFG_DEV int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
LFG_DEV int wave_sum_i(int v)
{
    v += dpp_i<0xB1>(v);            // quad_perm [1,0,3,2]
    v += dpp_i<0x4E>(v);            // quad_perm [2,3,0,1]
    v += dpp_i<0x141>(v);           // row_half_mirror
    v += dpp_i<0x140>(v);           // row_mirror: every lane of a row holds the row's sum
    return rl_i(v, 0) + rl_i(v, 16) + rl_i(v, 32) + rl_i(v, 48);
}
// Maximum / minimum of a double over the wave: four DPP exchanges inside the rows of 16 lanes (lane pairs, pairs of pairs,
// mirrored halves, mirrored rows: afterwards every lane of a row holds the row's value), then the four rows through scalar
// registers.  (The butterfly of six __shfl_xor is twelve trips through the LDS crossbar per call; region2rect makes four calls
// per region and most regions are a dozen pixels.)  Order free: the operands are finite.
template <int CTRL>
LFG_DEV double dpp_d(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
LFG_DEV double wave_max_d(double v)
{
    double o;
    o = dpp_d<0xB1>(v); v = o > v ? o : v;           // quad_perm [1,0,3,2]
    o = dpp_d<0x4E>(v); v = o > v ? o : v;           // quad_perm [2,3,0,1]
    o = dpp_d<0x141>(v); v = o > v ? o : v;          // row_half_mirror
    o = dpp_d<0x140>(v); v = o > v ? o : v;          // row_mirror
    const double r0 = rl_d(v, 0), r1 = rl_d(v, 16), r2 = rl_d(v, 32), r3 = rl_d(v, 48);
    const double a = r1 > r0 ? r1 : r0, b = r3 > r2 ? r3 : r2;
    return b > a ? b : a;
}
LFG_DEV double wave_min_d(double v)
{
    double o;
    o = dpp_d<0xB1>(v); v = o < v ? o : v;
    o = dpp_d<0x4E>(v); v = o < v ? o : v;
    o = dpp_d<0x141>(v); v = o < v ? o : v;
    o = dpp_d<0x140>(v); v = o < v ? o : v;
    const double r0 = rl_d(v, 0), r1 = rl_d(v, 16), r2 = rl_d(v, 32), r3 = rl_d(v, 48);
    const double a = r1 < r0 ? r1 : r0, b = r3 < r2 ? r3 : r2;
    return b < a ? b : a;
}
LFG_DEV int wave_max_i(int v)
{
    int o;
    o = dpp_i<0xB1>(v); v = o > v ? o : v;
    o = dpp_i<0x4E>(v); v = o > v ? o : v;
    o = dpp_i<0x141>(v); v = o > v ? o : v;
    o = dpp_i<0x140>(v); v = o > v ? o : v;
    const int a = max(rl_i(v, 0), rl_i(v, 16)), b = max(rl_i(v, 32), rl_i(v, 48));
    return max(a, b);
}
LFG_DEV void mem_fence() { __threadfence_block(); }
// A condition that is the same in every lane, said so: computed from values that live in vector registers (a double compared with a
// double) it counts as divergent for the compiler, and a loop that can be left through it becomes a loop over an exec mask -- every value
// it carries (the region size, the phase) moves to vector registers, every inner loop bound becomes a vector compare, every exit a chain
// of mask operations (13 cycles per dependent scalar instruction for a wave alone on its SIMD, tools/probe/lone_wave_issue.hip).
LFG_DEV bool uni(bool b) { return __builtin_amdgcn_readfirstlane((int)b) != 0; }
LFG_DEV int uni_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
#else
LFG_DEV bool uni(bool b) { return b; }
LFG_DEV int uni_i(int v) { return v; }
LFG_DEV int wave_max_i(int v) { return v; }
LFG_DEV int lane_id() { return 0; }
LFG_DEV int rl_i(int v, int) { return v; }
LFG_DEV float rl_f(float v, int) { return v; }
LFG_DEV double rl_d(double v, int) { return v; }
LFG_DEV int wave_sum_i(int v) { return v; }
LFG_DEV double wave_max_d(double v) { return v; }
LFG_DEV double wave_min_d(double v) { return v; }
LFG_DEV void mem_fence() {}
#endif

struct Rect { double x1, y1, x2, y2, width, x, y, theta, dx, dy, prec, p; };

// Diagnostic build only (-DLFG_STAMPS): cycle totals per phase, never enabled in the product.
#if defined(LFG_STAMPS) && !defined(LF_HOST_SIM)
#define LFG_T0 __builtin_amdgcn_sched_barrier(0); unsigned long long _t0 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
#define LFG_T1(c, k) { __builtin_amdgcn_sched_barrier(0); unsigned long long _t1 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); (c).stamps[k] += _t1 - _t0; _t0 = _t1; }
#define LFG_CNT(c, k, v) { (c).stamps[k] += (unsigned long long)(v); }
#else
#define LFG_CNT(c, k, v)
#define LFG_T0
#define LFG_T1(c, k)
#endif

// All per-pixel data live in COMPACT arrays indexed by e = raster rank among the problem's
// defined pixels (written by k_lsd_order); the dense scaled-image planes are never touched here.
// Deferred evaluation (GPU build).  Once a region has its final rectangle, what is left -- rect_improve's NFA search and
// the emission of the line -- reads only the rectangle and the problem's immutable arrays and changes no USED bit, so it
// does not have to happen before the next region is grown: the growing wave pushes (rectangle, seed tag) into a small
// ring in LDS and goes on to the next seed; waves of the workgroup that have run out of components pop and evaluate.
// Lines carry their seed's tag, so the order in which they are emitted does not matter (k_lsd_grow.hip restores it).
// Nobody ever waits on the ring: a push that finds it full (or the slot not yet copied out) evaluates in place, a pop
// that finds it empty returns; the only spin is a consumer waiting for the handful of stores of the producer whose
// ticket it has claimed.
#ifndef LFG_QN
#define LFG_QN 16
#endif
// LFG_EVAL_KERNEL=1 (round 3, measured and NOT the default): the evaluation in a kernel of its own -- the growing waves
// append (rectangle, tag) to the problem's PENDING list in HBM and k_lsd_eval (k_lsd_grow.hip), launched behind the
// growing kernel, evaluates the list with every wave of its workgroup working, so no wave sits through a problem's
// growth waiting for regions.  Same-call A/B on configs[1] (3 x 3 alternating runs): six batches in flight 139.7 k ->
// 139.3 k frames/s (+-0: a fifth fewer wave-slot-cycles per problem buy nothing, i.e. the pipelined rate is NOT bound by
// wave slots), one batch alone 2.77 -> 3.28 ms for growing + evaluation (the evaluation, 0.45 ms = its slowest wave's
// eleven regions at ~100 k cycles each, no longer hides under the growth).  Kept as a build option.
#ifndef LFG_EVAL_KERNEL
#define LFG_EVAL_KERNEL 0
#elif LFG_EVAL_KERNEL
#error "LFG_EVAL_KERNEL=1 fails a parity test since round 4 (tools/experiments/README.md): not buildable until that is fixed"
#endif
#ifndef LFG_EVAL_QUEUE
#define LFG_EVAL_QUEUE (LFG_EVAL_KERNEL ? 0 : 1)
#endif
struct EvalQueue {
    int tail, head;            // tickets reserved by producers / claimed by consumers
    int growers;               // waves of the workgroup that may still push
    int seq[LFG_QN];           // 0: slot free; t + 1: the item of ticket t is complete
    int tag[LFG_QN];
    double rec[LFG_QN][12];
};

struct Ctx {
    EvalQueue* q;             // LDS  deferred-evaluation ring of the workgroup, or nullptr (evaluate in place)
    int W, H;
    const int* rows;          // LDS  [H+1]  first entry of each row; rows[H] = number of defined pixels
    const uint16_t* lxs;      // LDS  x coordinate of entries e < def_lds (sorted inside a row)
    const uint32_t* gxy;      // HBM  y<<16|x of every entry
    int def_lds;              // -1: BITMAP form -- no row lists; the defined pixels as a bit plane + a count per 64-bit word:
    const unsigned long long* bits64;   // LDS  [H * W / 64 + 1] bit (y * W + x) set <=> the pixel's gradient is defined
    const uint16_t* pref;               // LDS  [H * W / 128 + 1] defined pixels in front of a PAIR of words = the compact index of its first one
    const float* deg;         // HBM  level-line angle, degrees (OpenCV fastAtan2 output)
    const double* mod;        // HBM  gradient magnitude
    const double* cs;         // HBM  cos((double)(float)angle_rad) -- device build: INTERLEAVED with sn, entry e at cs[2 e], cs[2 e + 1]
    const double* sn;         //      (one 16-byte load per candidate and one cache line instead of two; cs_sn() below)
    const float* sd;          // HBM  (float) cos, (float) sin of the angle as a double, pairs: what a region starting at the entry sums from (or null)
    uint32_t* usedc;          // LDS  USED bit per entry e < used_lds
    uint32_t* gused;          // HBM  USED bits of the remaining entries
    int used_lds;
    uint32_t* lreg;           // LDS  region list (y<<16|x), first reg_lds points
    uint32_t* greg;           // HBM  rest of the region list
    int reg_lds;
    double log_nt, log_eps, density_th, prec, p, scale;
    int min_reg_size, refine;
    // Connected components (k_lsd_label): region growing only ever moves along 8-adjacent DEFINED pixels and the
    // USED flags of other components are never consulted, so every connected component of the defined-pixel graph is
    // an independent sub-problem whose seeds are the global seed list restricted to it.  A wave works on ONE component
    // (label == root), several waves of a workgroup share the problem; lines carry their seed's position in the
    // global list (tag) and are put back into the sequential order afterwards.  label == nullptr: the whole problem.
    const uint16_t* label;    // HBM  component root of every entry
    int root;
    int* tags;                // seed position of every emitted line (HBM), or nullptr
    int* line_count;          // shared line counter (GPU: LDS, bumped atomically by the workgroup's waves)
    double* pend_rec;         // HBM  pending list of the problem: 12 doubles per finished region (LFG_EVAL_KERNEL) ...
    int* pend_tag;            // HBM  ... and its seed's position
    int* pend_n;              // LDS  entries appended so far (may pass pend_cap: the evaluation kernel reports it)
    int pend_cap;
#if defined(LFG_STAMPS) && !defined(LF_HOST_SIM)
    mutable unsigned long long stamps[32];  // 0 seed scan, 1 grow, 2 rect, 3 refine, 4 nfa scan, 5 nfa math, 6 emit, 7 nfa calls/px,
                                            // 8 seed fetch, 9 regions, 10 region points, 11 grow batches
#endif
};

LFG_DEV uint32_t reg_get(const Ctx& c, int i) { return i < c.reg_lds ? c.lreg[i] : c.greg[i]; }   // any lane, any index
LFG_DEV void reg_set(const Ctx& c, int i, uint32_t v)
{
    if (lane_id() == 0) { if (i < c.reg_lds) c.lreg[i] = v; else c.greg[i] = v; }
}
#ifndef LF_HOST_SIM
// LO = true: the caller knows the index is below reg_lds (region_grow decides per batch, wave-uniformly): a plain LDS access.  The
// generic forms above select between an LDS and a global address and compile to FLAT accesses -- the long way to LDS, and every
// wait behind one waits on both memory counters: the store of an accepted pixel made the ordered float sums behind it wait for
// the store to complete (s_waitcnt vmcnt(0) lgkmcnt(0) in the loop).
template <bool LO> LFG_DEV uint32_t reg_get_t(const Ctx& c, int i)
{
    if (LO) return ((__attribute__((address_space(3))) uint32_t*)c.lreg)[i];
    return reg_get(c, i);
}
template <bool LO> LFG_DEV void reg_put_t(const Ctx& c, int i, uint32_t v)       // the calling lane's slot
{
    if (LO) ((__attribute__((address_space(3))) uint32_t*)c.lreg)[i] = v;
    else if (i < c.reg_lds) c.lreg[i] = v; else c.greg[i] = v;
}
#endif
// cosine and sine of entry e's level-line angle
LFG_DEV void cs_sn(const Ctx& c, int e, double& ck, double& sk)
{
#ifndef LF_HOST_SIM
    const double2 v = *reinterpret_cast<const double2*>(c.cs + 2 * (size_t)e);
    ck = v.x; sk = v.y;
#else
    ck = c.cs[e]; sk = c.sn[e];
#endif
}
// (the angle in the same record as well -- 32 bytes per entry: cos, sin, angle, padding -- measured slower: kernel alone 1.90 -> 1.99
// ms, camera frames - 1.7 %; neighbouring entries of a row then share a cache line half as often)
LFG_DEV int xs_get(const Ctx& c, int e) { return e < c.def_lds ? (int)c.lxs[e] : (int)(c.gxy[e] & 0xffffu); }

// BITMAP form (round 4; k_lsd_grow.hip chooses it on busy content): a compact index is a RANK -- the number of defined pixels in
// front of a raster position -- so every lookup is two LDS reads and a popcount instead of a binary search in a row list, whatever
// the problem's size (problems beyond the LDS slice used to search their rows in global memory).
struct alignas(16) BmPair { unsigned long long x, y; };      // two words of the plane, read together
LFG_DEV bool bm_mode(const Ctx& c) { return c.def_lds == -1; }
LFG_DEV int rank_before(const Ctx& c, int pos)
{
    // one 16-byte read for the pair of words, one 2-byte read for its count
    const int w = pos >> 6;
    const BmPair v = reinterpret_cast<const BmPair*>(c.bits64)[w >> 1];
    const unsigned long long cur = (w & 1) ? v.y : v.x;
    return (int)c.pref[w >> 1] + ((w & 1) ? __builtin_popcountll(v.x) : 0) + __builtin_popcountll(cur & ((1ull << (pos & 63)) - 1ull));
}

// entry of pixel (x, y), or -1 when its gradient is undefined: binary search in row y's sorted list
LFG_DEV int find_e(const Ctx& c, int x, int y)
{
    if (bm_mode(c)) {
        const int pos = y * c.W + x, b = pos & 63;
        // both reads at once and no branch: the count is read whether the pixel turns out to be defined or not (a test of the bit
        // in front of it put a second LDS round trip on the growing wave's chain)
        const int wi = pos >> 6;
        const BmPair v = reinterpret_cast<const BmPair*>(c.bits64)[wi >> 1];
        const int pf = (int)c.pref[wi >> 1];
        const unsigned long long w = (wi & 1) ? v.y : v.x;
        const int e = pf + ((wi & 1) ? __builtin_popcountll(v.x) : 0) + __builtin_popcountll(w & ((1ull << b) - 1ull));
        return ((w >> b) & 1ull) ? e : -1;
    }
    int lo = c.rows[y];
    const int end = c.rows[y + 1];
    int hi = end;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (xs_get(c, mid) < x) lo = mid + 1; else hi = mid;
    }
    return (lo < end && xs_get(c, lo) == x) ? lo : -1;
}

#ifndef LF_HOST_SIM
// The USED bits' LDS part and HBM part are reached through pointers of their own address spaces.  Written as
// `e < n ? lds[..] : hbm[..]` on generic pointers the compiler selects the ADDRESS and issues ONE flat access, which takes the
// long way to LDS and, counted as a vector-memory operation too, makes the wave wait for every global load in flight.
// k_lsd_grow<false> (every problem of the 640x480 geometries) sets used_lds to "infinite": the HBM halves fold away and these
// are plain LDS operations -- kernel alone 2.79 -> 2.61 ms, +3 % frames/s (same-call A/B; the same treatment of the region
// lists, whose HBM tail is real, measured slower: the per-lane branch costs more than the flat access).
typedef __attribute__((address_space(3))) uint32_t lds_u32;
LFG_DEV lds_u32* as_lds(uint32_t* p) { return (lds_u32*)p; }
// (`e < 0x7fffffff` does not fold -- e might be INT_MAX for all the compiler knows -- and left a branch and the global half of
// every USED operation in the all-in-LDS kernels; the equality does)
LFG_DEV bool used_in_lds(const Ctx& c, int e) { return c.used_lds == 0x7fffffff || e < c.used_lds; }
LFG_DEV bool used_get(const Ctx& c, int e)
{
    uint32_t w;
    if (used_in_lds(c, e)) w = as_lds(c.usedc)[e >> 5]; else w = c.gused[e >> 5];
    return (w >> (e & 31)) & 1u;
}
// atomics without return value: no read-modify-write round trip on the sequential path
LFG_DEV void used_or(const Ctx& c, int e)                // the calling lane's entry
{
    if (used_in_lds(c, e)) __hip_atomic_fetch_or(as_lds(c.usedc) + (e >> 5), 1u << (e & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else atomicOr(&c.gused[e >> 5], 1u << (e & 31));
}
LFG_DEV void used_set(const Ctx& c, int e) { if (lane_id() == 0) used_or(c, e); }
LFG_DEV void used_and(const Ctx& c, int e)               // the calling lane's entry
{
    if (used_in_lds(c, e)) __hip_atomic_fetch_and(as_lds(c.usedc) + (e >> 5), ~(1u << (e & 31)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else atomicAnd(&c.gused[e >> 5], ~(1u << (e & 31)));
}
LFG_DEV void used_clr(const Ctx& c, int e)
{
    if (lane_id() == 0) {
        if (used_in_lds(c, e)) __hip_atomic_fetch_and(as_lds(c.usedc) + (e >> 5), ~(1u << (e & 31)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else atomicAnd(&c.gused[e >> 5], ~(1u << (e & 31)));
    }
}
#else
LFG_DEV bool used_get(const Ctx& c, int e)
{
    const uint32_t w = e < c.used_lds ? c.usedc[e >> 5] : c.gused[e >> 5];
    return (w >> (e & 31)) & 1u;
}
LFG_DEV void used_set(const Ctx& c, int e) { if (e < c.used_lds) c.usedc[e >> 5] |= 1u << (e & 31); else c.gused[e >> 5] |= 1u << (e & 31); }
LFG_DEV void used_clr(const Ctx& c, int e) { if (e < c.used_lds) c.usedc[e >> 5] &= ~(1u << (e & 31)); else c.gused[e >> 5] &= ~(1u << (e & 31)); }
#endif

// sine and cosine of an angle through the deterministic routine, out of line for the same reason as nfa_eval below
struct SinCos { double s, c; };
#ifndef LF_HOST_SIM
__device__ __noinline__
#else
static
#endif
SinCos sincos_eval(double x)
{
    SinCos r;
    dm::dsincos(x, r.s, r.c);
    return r;
}

LFG_DEV double angle_of(float deg) { return deg == NOTDEF_F ? NOTDEF_D : (double)deg * DEG2RAD; }

LFG_DEV bool aligned_val(double a, double theta, double prec)
{
#ifndef LF_HOST_SIM
    // the same values without a branch (three nested ifs were three exec-mask regions on the growing wave's chain)
    double n_theta = theta - a;
    n_theta = n_theta < 0 ? -n_theta : n_theta;
    double t2 = n_theta - M_2__PI_;
    t2 = t2 < 0 ? -t2 : t2;
    n_theta = n_theta > M_3_2_PI_ ? t2 : n_theta;
    return (a != NOTDEF_D) & (n_theta <= prec);
#else
    if (a == NOTDEF_D) return false;
    double n_theta = theta - a;
    if (n_theta < 0) n_theta = -n_theta;
    if (n_theta > M_3_2_PI_) {
        n_theta -= M_2__PI_;
        if (n_theta < 0) n_theta = -n_theta;
    }
    return n_theta <= prec;
#endif
}

LFG_DEV double angle_diff_signed(double a, double b)
{
    double diff = a - b;
    while (uni(diff <= -PI_)) diff += M_2__PI_;          // (every caller's operands are wave-uniform)
    while (uni(diff > PI_)) diff -= M_2__PI_;
    return diff;
}
LFG_DEV double dabs(double v) { return v < 0 ? -v : v; }
LFG_DEV double dist_sq(double x1, double y1, double x2, double y2) { return (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1); }
LFG_DEV double dist_(double x1, double y1, double x2, double y2) { return dm::dsqrt(dist_sq(x1, y1, x2, y2)); }

LFG_DEV bool double_equal(double a, double b)
{
    if (a == b) return true;
    double abs_diff = dabs(a - b);
    double aa = dabs(a), bb = dabs(b);
    double abs_max = (aa > bb) ? aa : bb;
    const double DBL_MIN_ = 2.2250738585072014e-308, DBL_EPS_ = 2.2204460492503131e-16;
    if (abs_max < DBL_MIN_) abs_max = DBL_MIN_;
    return (abs_diff / abs_max) <= (100.0 * DBL_EPS_);
}

// ------------------------------------------------------------------ region_grow
// seed_e: compact entry of the seed pixel (sx, sy)
LFG_DEV void region_grow(const Ctx& c, int sx, int sy, int seed_e, int& reg_size, double& reg_angle, double prec, float pre_deg = -4096.f)
{
    const int W = c.W, H = c.H;
    const int lane = lane_id();
    int n = 1;
#ifndef LF_HOST_SIM
    if (lane == 0) {                                          // (entry 0 is in the LDS part whenever there is one: a plain LDS store, see reg_get_t)
        if (c.reg_lds > 0) reg_put_t<true>(c, 0, ((uint32_t)sy << 16) | (uint32_t)sx);
        else c.greg[0] = ((uint32_t)sy << 16) | (uint32_t)sx;
    }
#else
    reg_set(c, 0, ((uint32_t)sy << 16) | (uint32_t)sx);
#endif
    used_set(c, seed_e);
#ifdef LF_HOST_SIM
    reg_angle = angle_of(c.deg[seed_e]);
    const SinCos sc0 = sincos_eval(reg_angle);
    float sumdx = (float)sc0.c, sumdy = (float)sc0.s;
#endif
#ifndef LF_HOST_SIM
    // Frontier points are taken in list order, 7 at a time: lane = 9*slot + neighbour, so
    // ascending lane order IS the reference's visiting order (point i, then its 3x3 window in
    // raster order).  Each lane locates its neighbour in the row lists (LDS binary search) and
    // fetches angle / cos / sin from the compact arrays in one round trip.  The reference tests
    // neighbours one by one against the running region angle, which only changes when a pixel is
    // accepted; so "the next accepted pixel" is the lowest lane at or after the cursor whose
    // pixel is free and aligned under the CURRENT angle -- one vector evaluation + ballot per
    // accepted pixel instead of a 9-step scalar chain per point.
    LFG_T0
    const int slot = lane / 9, k9 = lane - slot * 9;      // lane constants
    const int ddx = (k9 % 3) - 1, ddy = (k9 / 3) - 1;
    // Seed window: the 9 x 7 pixels around the seed, one per lane (lane = 9*row + column), located and
    // fetched in ONE round trip when the region starts.  Most regions are a handful of pixels (noise
    // that fails min_reg_size, short dashes) and never leave it, so their later batches need neither the
    // row search nor another trip to the compact arrays: a neighbour inside the window is a lane
    // shuffle away.  Neighbours outside it take the per-lane search + fetch below, as before.
    const int wx0 = sx - 4, wy0 = sy - 3;
    int w_e = -1;
    float w_deg = 0.0f;
    double w_cs = 0.0, w_sn = 0.0;
    {
        const int x = wx0 + k9, y = wy0 + slot;
        if (lane < 63 && x >= 0 && x < W && y >= 0 && y < H) w_e = find_e(c, x, y);
    }
    // Two thirds of all seeds are ISOLATED: leftovers of earlier regions whose eight neighbours are all undefined or
    // already used.  Their region is the seed alone whatever the angles are, and (for min_reg_size > 1) it is thrown
    // away -- so it is recognised here from the row lists and the USED bits in LDS, before anything is fetched from
    // the compact arrays and before the seed's sine / cosine are worked out.
    // (the same mask is the candidate mask of the region's FIRST batch: the seed's eight neighbours, in the window's own lanes)
    unsigned long long firstm;
    {
        const int ax = k9 - 4, ay = slot - 3;
        const bool adj = lane < 63 && ax >= -1 && ax <= 1 && ay >= -1 && ay <= 1 && (ax != 0 || ay != 0);
        firstm = lfg_ballot(adj && w_e >= 0 && !used_get(c, w_e));                          // (measured: without the branch around the LDS read, + 2 %)
        if (c.min_reg_size > 1 && firstm == 0ull) {
            reg_size = 1;
            reg_angle = NOTDEF_D;
            LFG_CNT(c, 20, 1)
            return;
        }
    }
    // The seed's own angle comes with the seed list when detect() has it (pre_deg: fetched per chunk, beside the seeds' coordinates):
    // its sine and cosine are then worked out INLINE while the window's angles are on their way -- through the out-of-line routine the
    // call would first wait for every load in flight, and with the angle from the compact array there was a round trip in front of it
    const bool pre = __builtin_amdgcn_readfirstlane((int)(pre_deg != -4096.f)) != 0;
    SinCos sc0;
    float2 sd0 = make_float2(0.f, 0.f);
    const bool have_sd = pre && c.sd != nullptr;
    if (have_sd) {
        // k_lsd_grad has worked the seed's first sums out (LsdParams::c_sd): a load beside the window's instead of a double sine and
        // cosine inline under them -- a third of the 1 800 cycles a region's start took, 5 400 times per clutter problem
        reg_angle = angle_of(pre_deg);
        sd0 = *reinterpret_cast<const float2*>(c.sd + 2 * (size_t)seed_e);
        if (w_e >= 0) { w_deg = c.deg[w_e]; cs_sn(c, w_e, w_cs, w_sn); }
        sc0.c = 0; sc0.s = 0;
    } else if (pre) {
        reg_angle = angle_of(pre_deg);
        if (w_e >= 0) { w_deg = c.deg[w_e]; cs_sn(c, w_e, w_cs, w_sn); }
        dm::dsincos(reg_angle, sc0.s, sc0.c);
    } else {
        reg_angle = angle_of(c.deg[seed_e]);
        if (w_e >= 0) { w_deg = c.deg[w_e]; cs_sn(c, w_e, w_cs, w_sn); }
        sc0 = sincos_eval(reg_angle);
    }
    float sumdx = have_sd ? sd0.x : (float)sc0.c, sumdy = have_sd ? sd0.y : (float)sc0.s;
    LFG_T1(c, 16)
    const float precf = (float)prec;
    const float EPSF = 0.0043633f;                        // 0.25 degree
    const bool bulk_ok = __builtin_amdgcn_readfirstlane((int)(precf < 0.7f && precf > 4.f * EPSF)) != 0;     // (uniform, said explicitly: see the accept loop)
    const float tn_cone = 0.5f * (precf - EPSF), tf_cone = precf + 0.5f * (precf - EPSF) + EPSF;
    for (int i = 0; i < n;) {
      // one batch; LO: every list index it touches (i .. n + 63) is in the LDS part of the list -- nearly always (see reg_get_t)
      auto batch = [&](auto lo_tag, auto first_tag) {
        constexpr bool LO = decltype(lo_tag)::value;
        // FIRST: the region's first batch -- one frontier point, the seed, whose eight neighbours ARE window lanes (lane = 9 * row +
        // column: ascending lane order is their raster order, the reference's visiting order) and whose candidate mask the
        // isolated-seed test has just worked out: no list read, no shuffles, no second look at the USED bits
        constexpr bool FIRST = decltype(first_tag)::value;
        int m, e, xx, yy;
        float dg;
        double ck, sk;
        unsigned long long candm;
        if constexpr (FIRST) {
            m = 1; e = w_e; dg = w_deg; ck = w_cs; sk = w_sn;
            xx = wx0 + k9; yy = wy0 + slot;
            candm = firstm;
        } else {
        m = n - i < 7 ? n - i : 7;
        const bool lv = slot < m;
        const uint32_t pkl = lv ? reg_get_t<LO>(c, i + slot) : 0u;
        const int pxl = (int)(pkl & 0xffffu), pyl = (int)(pkl >> 16);
        xx = pxl + ddx; yy = pyl + ddy;
        const bool inb = lv && xx >= 0 && xx < W && yy >= 0 && yy < H;
        const int rx = xx - wx0, ry = yy - wy0;
        const bool inwin = lv && rx >= 0 && rx < 9 && ry >= 0 && ry < 7;
        const int widx = inwin ? ry * 9 + rx : 0;
        e = __shfl(w_e, widx);
        dg = __shfl(w_deg, widx);
        ck = __shfl(w_cs, widx); sk = __shfl(w_sn, widx);
        LFG_T1(c, 12)
        LFG_CNT(c, 15, lfg_ballot(inb && !inwin) != 0ull)
        if (!inwin) {
            e = inb ? find_e(c, xx, yy) : -1;
            if (e >= 0) { dg = c.deg[e]; cs_sn(c, e, ck, sk); }      // fetched whether USED or not: the USED test below is an LDS round trip that need not sit in front of the trip to the compact arrays
        }
        // the candidates -- defined and free at batch start -- as a LANE MASK (wave-uniform integers from here on: a per-lane bool
        // that survives a branch comes back through a 0 / 1 register and a compare every time it is balloted, three instructions
        // and a vector -> scalar dependency, eight times per span)
        candm = lfg_ballot((e >= 0) & !used_get(c, e < 0 ? 0 : e));      // (no branch around the LDS read)
        }
        const double a = (double)dg * DEG2RAD;
        LFG_T1(c, 13)
        // The reference tests the candidates one by one, in lane order, each against the running region angle,
        // which moves with every accepted pixel.  Most decisions do not depend on that order: the running angle
        // can only move a little while a handful of nearly parallel unit vectors join a sum of many, so a
        // candidate well inside the tolerance is accepted and one well outside is rejected WHATEVER the order.
        // Per span (cursor .. first undecided lane) the candidates are therefore split three ways against the
        // angle R at the span's start, with d = |angle - R| folded to [0, pi]:
        //   near  d <= Tn        accepted, all at once (list slots by prefix count, USED bits by per-lane atomics,
        //                        the two float sums by a short ordered chain -- no arc tangent per pixel)
        //   far   d >  Tf        rejected
        //   mid   otherwise      undecided: the span ends here, the lane gets the reference's own comparison
        //                        under the exact angle of that moment, and the rest is classified afresh.
        // Two valid (Tn, Tf) pairs, the wider near band is taken (EPS covers fastAtan2's 0.011 degree error, the
        // float sums and this float classification, with a margin of 20x):
        //   cone       Tn = (prec - EPS) / 2, Tf = prec + Tn + EPS: while only near lanes join, the sum stays
        //              inside the cone R +- Tn (a cone narrower than pi is closed under addition), so a near lane is
        //              at most 2 Tn + EPS = prec away from any intermediate angle, a far lane more than prec;
        //   magnitude  delta = (pi/2) m / max(|sumdx|, |sumdy|) >= asin(m / |S|) bounds how far m unit vectors can
        //              turn the sum S: Tn = prec - delta - EPS, Tf = prec + delta + EPS, m = candidates within prec.
        // Duplicates (the same pixel offered by several frontier points of this batch) keep the reference's rule
        // "first offer wins": inside a span the near lanes are walked in order and every new pixel strikes its later
        // offers; across spans the USED bits decide, re-read after every bulk accept.
        const float af = (float)a;
        const uint32_t key = ((uint32_t)yy << 16) | (uint32_t)xx;
        unsigned long long later = ~0ull;                 // lanes at or after the cursor
        bool added = false;
        for (;;) {
            const unsigned long long cb = candm & later;
            if (cb == 0ull) break;                        // nobody left to test
            float ad = af - (float)reg_angle;
            ad = ad < 0.f ? -ad : ad;
            if (ad > 3.14159265f) ad = 6.28318531f - ad;
            unsigned long long nearM = 0ull, midM = ~0ull;    // everything undecided = the reference's loop, lane by lane
            if (bulk_ok) {
                // division free: ad <= prec - EPS - (pi/2) m / smax  <=>  (ad - prec + EPS) smax + (pi/2) m <= 0
                const float hm = 1.5707964f * (float)(int)__popcll(lfg_ballot(ad <= precf) & cb);
                const float sx_ = sumdx < 0.f ? -sumdx : sumdx, sy_ = sumdy < 0.f ? -sumdy : sumdy;
                const float smax = sx_ > sy_ ? sx_ : sy_;
                // the pair with the wider near band, chosen by SELECTS: both pairs as thresholds (delta a shade above (pi/2) m / smax -- the
                // reciprocal is approximate; either way the slack is a millionth of what EPS allows for), no branch and no trip through a scalar
                // register in front of the ballots
                const float delta = hm * __builtin_amdgcn_rcpf(smax) * 1.000001f;
                const float tn_mag = precf - EPSF - delta;
                const bool mag = tn_mag > tn_cone;
                const float tn = mag ? tn_mag : tn_cone, tf = mag ? precf + EPSF + delta : tf_cone;
                nearM = lfg_ballot(ad <= tn);
                midM = ~nearM & lfg_ballot(ad <= tf);
            }
            const unsigned long long maskM = midM & cb;
            // the span = the lanes in front of the first undecided one (all of them when there is none): the bits below maskM's lowest
            const unsigned long long span = (maskM - 1ull) & ~maskM;
            const unsigned long long maskN = nearM & cb & span;
            LFG_CNT(c, 27, 1) LFG_CNT(c, 28, (maskN == 0ull && maskM != 0ull)) LFG_CNT(c, 30, (maskN & (maskN - 1ull)) != 0ull)
            if (maskN != 0ull) {
                // first offer wins inside the span: walk the near lanes in order, each new pixel strikes its later offers
                unsigned long long maskA = maskN;
                if (maskN & (maskN - 1ull)) {
                    for (unsigned long long mm = maskN; mm != 0ull;) {
                        const int j = __builtin_ctzll(mm);
                        const unsigned long long same = lfg_ballot(key == (uint32_t)rl_i((int)key, j)) & mm & ~(1ull << j);
                        maskA &= ~same;
                        mm &= ~(same | (1ull << j));
                    }
                }
                const bool acc = (maskA >> lane) & 1ull;
                // region-list slot = n + number of accepted lanes below me
                const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(maskA >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)maskA, 0u));
                if (acc) {
                    reg_put_t<LO>(c, n + below, key);
                    used_or(c, e);
                }
                n += __popcll(maskA);
                // the float sums take the accepted vectors in list order: (float)((double)sum + cos), one by one
                for (unsigned long long mm = maskA; mm != 0ull; mm &= mm - 1ull) {
                    const int j = __builtin_ctzll(mm);
                    sumdx = (float)((double)sumdx + rl_d(ck, j));
                    sumdy = (float)((double)sumdy + rl_d(sk, j));
                }
                reg_angle = (double)dm::fast_atan2_deg(sumdy, sumdx) * DEG2RAD;
                added = true;
                LFG_CNT(c, 18, __popcll(maskA))
            }
            if (maskM == 0ull) break;
            const int L = __builtin_ctzll(maskM);
            if (maskN != 0ull) {
                // later offers of the pixels just accepted are no longer candidates (measured: striking them by key in the loop
                // above instead of this trip to the USED bits made the kernel 1.5 % slower)
                mem_fence();
                candm &= ~lfg_ballot((e >= 0) & used_get(c, e < 0 ? 0 : e));
            }
            // the undecided lane: the reference's comparison under the angle of this moment
            // (measured: the same comparison on lane L's angle as a wave-uniform value -- readlane, scalar branches -- made the
            // kernel 15 % slower)
            // When lane L fails under the angle of this moment, so does every candidate behind it up to the first that passes (the angle
            // only moves when a pixel is accepted): the loop goes on AT that lane, one trip for all of them instead of one each (a third
            // of all spans on the refined, narrow-tolerance growths).  (`later` is settled before the branch: nothing of this is live
            // across the accept code)
            const bool hitL = [&]() {
                const unsigned long long hits = lfg_ballot(aligned_val(a, reg_angle, prec)) & candm;
                const unsigned long long behind = ~(maskM ^ (maskM - 1ull));     // the lanes behind the undecided one
                const bool hit = (hits >> L) & 1ull;
                const unsigned long long h2 = hits & behind;
                later = hit ? behind : ~((h2 & (0ull - h2)) - 1ull);            // miss: the lanes at or behind h2's lowest; none when h2 is empty
                return hit;
            }();
            if (hitL) {
                const int eL = rl_i(e, L);
                const int ay = rl_i(yy, L), ax = rl_i(xx, L);
                used_set(c, eL);
                if (lane == 0) reg_put_t<LO>(c, n, ((uint32_t)ay << 16) | (uint32_t)ax);
                ++n;
                sumdx = (float)((double)sumdx + rl_d(ck, L));
                sumdy = (float)((double)sumdy + rl_d(sk, L));
                reg_angle = (double)dm::fast_atan2_deg(sumdy, sumdx) * DEG2RAD;
                candm &= ~lfg_ballot(e == eL);            // the same pixel seen from a later point is now USED
                added = true;
                LFG_CNT(c, 19, 1)
            } else { LFG_CNT(c, 29, 1) }
        }
        if (added) mem_fence();
        LFG_T1(c, 14)
        LFG_CNT(c, 11, 1)
        i += m;
      };
      if (i == 0 && 65 <= c.reg_lds) batch(std::true_type{}, std::true_type{});
      else if (n + 64 <= c.reg_lds) batch(std::true_type{}, std::false_type{});
      else batch(std::false_type{}, std::false_type{});
    }
#else
    for (int i = 0; i < n; ++i) {
        const uint32_t pk = reg_get(c, i);
        const int px = (int)(pk & 0xffffu), py = (int)(pk >> 16);
        for (int k = 0; k < 9; ++k) {
            const int xx = px + (k % 3) - 1, yy = py + (k / 3) - 1;
            if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
            const int e = find_e(c, xx, yy);
            if (e < 0 || used_get(c, e)) continue;
            const double ak = (double)c.deg[e] * DEG2RAD;
            if (aligned_val(ak, reg_angle, prec)) {
                used_set(c, e);
                reg_set(c, n, ((uint32_t)yy << 16) | (uint32_t)xx);
                ++n;
                sumdx = (float)((double)sumdx + c.cs[e]);
                sumdy = (float)((double)sumdy + c.sn[e]);
                reg_angle = (double)dm::fast_atan2_deg(sumdy, sumdx) * DEG2RAD;
            }
        }
    }
#endif
    reg_size = n;
}

#ifndef LF_HOST_SIM
// Three ORDERED f64 sums at once: s0 += t0[0] + t0[1] + ..., s1 += t1[..], s2 += t2[..] (left to right, term j held by lane j, cnt <= 64
// terms), the running sums in lanes 0, 1, 2 of `acc`.  The terms are first dealt out so that lane 3 j + L holds term j of sum L (21 terms
// per deal: three ds_bpermute per sum), then step j is ONE v_add_f64 whose operand lanes 0 .. 2 fetch from lanes 3 j .. 3 j + 2 (a
// ds_bpermute with the lane's own address + a constant: the LDS crossbar, no LDS memory, nothing on the chain but the addition).  Terms
// past cnt are -0.0, which x + (-0.0) == x leaves alone for every x, so the steps come in straight-line groups of seven without a count
// in the loop.  Before: v_readlane x 2 per term and sum, a scalar counter, a compare and a branch per term -- 12 instructions and ~100
// cycles per region point and pass for a wave alone on its SIMD; region2rect was 13 % of the longest growing wave's time.
LFG_DEV double bperm_d(int addr, double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_ds_bpermute(addr, (int)(b & 0xffffffffll));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, (int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
LFG_DEV void ordered_sums3(double& acc, double t0, double t1, double t2, int cnt)
{
    const int lane = lane_id();
    const int third = lane / 3, rem = lane - 3 * third;
    if (lane >= cnt) { t0 = -0.0; t1 = -0.0; t2 = -0.0; }
    for (int sb = 0; sb * 21 < cnt; ++sb) {                  // (cnt is wave-uniform: scalar loops)
        const int p = third + 21 * sb;                        // the term this lane is dealt
        const double a0 = bperm_d(p << 2, t0), a1 = bperm_d(p << 2, t1), a2 = bperm_d(p << 2, t2);
        double T = rem == 0 ? a0 : (rem == 1 ? a1 : a2);
        if (p > 63) T = -0.0;                                 // (the address wraps: term 64 is not term 0)
        for (int j0 = 0; j0 < 21; j0 += 7) {
            if (21 * sb + j0 >= cnt) break;
            // (all seven fetches in flight, then the seven additions: left to itself the compiler waits for each fetch in turn)
            double u[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) u[j] = bperm_d((lane + 3 * (j0 + j)) << 2, T);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 7; ++j) acc += u[j];
        }
    }
}
#endif

// ------------------------------------------------------------------ region2rect
// LO (device build): the whole region list is in its LDS part (reg_size <= reg_lds, nearly always): plain LDS reads (reg_get_t).
// The ordered f64 sums take their terms from lane registers in list order; the products that go into them do not depend on the
// order and are formed by all lanes at once (same operands, same operations, no contraction: the same bits) -- the sequential part
// of a point is three additions.
template <bool LO>
LFG_DEV void region2rect_t(const Ctx& c, int reg_size, double reg_angle, double prec, double p, Rect& rec)
{
    const int lane = lane_id();
#ifndef LF_HOST_SIM
    auto list_at = [&](int i) { return reg_get_t<LO>(c, i); };
#else
    auto list_at = [&](int i) { return reg_get(c, i); };
#endif
    double x = 0, y = 0, sum = 0;
    // points and weights of the first 64 region points stay in lane registers for the second pass
    // (most regions are shorter than that: one row search + one trip to the magnitudes instead of two)
    uint32_t pk0 = 0u;
    double w0 = 0.0;
#ifndef LF_HOST_SIM
    double acc = 0.0;                                        // lanes 0, 1, 2: the three running sums (ordered_sums3)
#endif
    for (int base = 0; base < reg_size; base += LFG_NL) {
        const int i = base + lane;
        const bool v = i < reg_size;
        const uint32_t pk = v ? list_at(i) : 0u;
        const double w = v ? c.mod[find_e(c, (int)(pk & 0xffffu), (int)(pk >> 16))] : 0.0;
        if (base == 0) { pk0 = pk; w0 = w; }
        const double xw = (double)(int)(pk & 0xffffu) * w, yw = (double)(int)(pk >> 16) * w;
        const int cnt = reg_size - base < LFG_NL ? reg_size - base : LFG_NL;
#ifndef LF_HOST_SIM
        ordered_sums3(acc, xw, yw, w, cnt);
#else
        for (int j = 0; j < cnt; ++j) {
            x += rl_d(xw, j);
            y += rl_d(yw, j);
            sum += rl_d(w, j);
        }
#endif
    }
#ifndef LF_HOST_SIM
    x = rl_d(acc, 0); y = rl_d(acc, 1); sum = rl_d(acc, 2);
    acc = 0.0;
#endif
    x /= sum;
    y /= sum;
    // get_theta
    double Ixx = 0.0, Iyy = 0.0, Ixy = 0.0;
    for (int base = 0; base < reg_size; base += LFG_NL) {
        const int i = base + lane;
        const bool v = i < reg_size;
        uint32_t pk = pk0;
        double w = w0;
        if (base != 0) {
            pk = v ? list_at(i) : 0u;
            w = v ? c.mod[find_e(c, (int)(pk & 0xffffu), (int)(pk >> 16))] : 0.0;
        }
        const double ddx = (double)(int)(pk & 0xffffu) - x;
        const double ddy = (double)(int)(pk >> 16) - y;
        const double tyy = ddy * ddy * w, txx = ddx * ddx * w, txy = ddx * ddy * w;
        const int cnt = reg_size - base < LFG_NL ? reg_size - base : LFG_NL;
#ifndef LF_HOST_SIM
        ordered_sums3(acc, tyy, txx, -txy, cnt);             // (Ixy -= t is Ixy += -t, bit for bit)
#else
        for (int j = 0; j < cnt; ++j) {
            Ixx += rl_d(tyy, j);
            Iyy += rl_d(txx, j);
            Ixy -= rl_d(txy, j);
        }
#endif
    }
#ifndef LF_HOST_SIM
    Ixx = rl_d(acc, 0); Iyy = rl_d(acc, 1); Ixy = rl_d(acc, 2);
#endif
    const double lambda = 0.5 * (Ixx + Iyy - dm::dsqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
    double theta = (dabs(Ixx) > dabs(Iyy)) ? (double)dm::fast_atan2_deg((float)(lambda - Ixx), (float)Ixy)
                                           : (double)dm::fast_atan2_deg((float)Ixy, (float)(lambda - Iyy));
    theta *= DEG2RAD;
    if (dabs(angle_diff_signed(theta, reg_angle)) > prec) theta += PI_;
    const SinCos sct = sincos_eval(theta);
    const double dx = sct.c, dy = sct.s;
    // extents: max/min are order independent -> lane-parallel
    double l_min = 0, l_max = 0, w_min = 0, w_max = 0;
    for (int i = lane; i < reg_size; i += LFG_NL) {
        const uint32_t pk = list_at(i);
        const double regdx = (double)(int)(pk & 0xffffu) - x;
        const double regdy = (double)(int)(pk >> 16) - y;
        const double l = regdx * dx + regdy * dy;
        const double w = -regdx * dy + regdy * dx;
        if (l > l_max) l_max = l; else if (l < l_min) l_min = l;
        if (w > w_max) w_max = w; else if (w < w_min) w_min = w;
    }
    l_max = wave_max_d(l_max); l_min = wave_min_d(l_min);
    w_max = wave_max_d(w_max); w_min = wave_min_d(w_min);
    rec.x1 = x + l_min * dx;
    rec.y1 = y + l_min * dy;
    rec.x2 = x + l_max * dx;
    rec.y2 = y + l_max * dy;
    rec.width = w_max - w_min;
    rec.x = x; rec.y = y; rec.theta = theta; rec.dx = dx; rec.dy = dy;
    rec.prec = prec; rec.p = p;
    if (rec.width < 1.0) rec.width = 1.0;
}
LFG_DEV void region2rect(const Ctx& c, int reg_size, double reg_angle, double prec, double p, Rect& rec)
{
#ifndef LF_HOST_SIM
    if (reg_size <= c.reg_lds) region2rect_t<true>(c, reg_size, reg_angle, prec, p, rec);
    else
#endif
        region2rect_t<false>(c, reg_size, reg_angle, prec, p, rec);
}

// ------------------------------------------------------------------ refine
// OpenCV's refine() / reduce_region_radius() in pieces: detect() below drives them from ONE loop so that region_grow and
// region2rect are instantiated once (they are the two largest inlined bodies; every copy is instruction-cache footprint).
//
// One radius-reduction step: shrink the radius by 0.75 and drop the region points beyond it (USED cleared).
#ifndef LF_HOST_SIM
LFG_DEV void reg_put(const Ctx& c, int i, uint32_t v) { if (i < c.reg_lds) c.lreg[i] = v; else c.greg[i] = v; }      // the calling lane's slot
// dense-from-sparse inside a wave: the lanes of `mask` hand `v` to lanes 0, 1, 2, ... in lane order (the other lanes' values
// land behind them); one ds_permute
LFG_DEV int wave_compact(int v, unsigned long long mask)
{
    const int lane = lane_id();
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    const bool set = (mask >> lane) & 1ull;
    const int dest = set ? __popcll(mask & below) : __popcll(mask) + __popcll(~mask & below);
    return __builtin_amdgcn_ds_permute(dest << 2, v);
}
#endif

LFG_DEV void reduce_radius_step(const Ctx& c, int& reg_size, double xc, double yc, double& radSq)
{
    radSq *= 0.75 * 0.75;
#ifndef LF_HOST_SIM
    // The reference walks the list once and overwrites every point beyond the radius with the list's current last point
    // (which is tested next).  What comes out: m = the points that stay; those of them in the first m places keep their
    // places; the places of the first m that fall free are taken, in ascending order, by the staying points from behind place
    // m, in DESCENDING order.  (Checked against the sequential loop on random lists.)  So: one pass that tests all points
    // and releases the dropped ones, lane-parallel, and one that pairs holes and fillers through two small queues in registers.
    const int n = reg_size, lane = lane_id();
    int m = 0;
    for (int base = 0; base < n; base += LFG_NL) {
        const int i = base + lane;
        const bool v = i < n;
        const uint32_t pk = v ? reg_get(c, i) : 0u;
        const int px = (int)(pk & 0xffffu), py = (int)(pk >> 16);
        const bool out = v && dist_sq(xc, yc, (double)px, (double)py) > radSq;
        if (out) used_and(c, find_e(c, px, py));
        m += __popcll(lfg_ballot(v && !out));
    }
    if (m == n) return;
    int hb = 0, fb = n;                 // front places scanned so far: [0, hb); back places scanned so far: [fb, n)
    int hq = 0, fq = 0, nh = 0, nf = 0; // lane j: the j-th waiting hole (a place) / the j-th waiting filler (a point)
    for (;;) {
        if (nh == 0) {
            if (hb >= m) break;         // no hole left (the fillers are used up with them)
            const int i = hb + lane;
            const bool v = i < m;
            const uint32_t pk = v ? reg_get(c, i) : 0u;
            const bool out = v && dist_sq(xc, yc, (double)(int)(pk & 0xffffu), (double)(int)(pk >> 16)) > radSq;
            const unsigned long long mk = lfg_ballot(out);
            hq = wave_compact(i, mk);
            nh = __popcll(mk);
            hb += LFG_NL;
            if (nh == 0) continue;
        }
        if (nf == 0) {
            if (fb <= m) break;                             // cannot happen (as many fillers as holes): never spin
            const int i = fb - 1 - lane;                    // descending with the lane
            const bool v = i >= m;
            const uint32_t pk = v ? reg_get(c, i) : 0u;
            const bool keep = v && !(dist_sq(xc, yc, (double)(int)(pk & 0xffffu), (double)(int)(pk >> 16)) > radSq);
            const unsigned long long mk = lfg_ballot(keep);
            fq = wave_compact((int)pk, mk);
            nf = __popcll(mk);
            fb -= LFG_NL;
            if (nf == 0) continue;
        }
        const int t = nh < nf ? nh : nf;
        if (lane < t) reg_put(c, hq, (uint32_t)fq);
        hq = __builtin_amdgcn_ds_bpermute(((lane + t) & 63) << 2, hq);
        fq = __builtin_amdgcn_ds_bpermute(((lane + t) & 63) << 2, fq);
        nh -= t; nf -= t;
    }
    mem_fence();
    reg_size = m;
#else
    for (int i = 0; i < reg_size; ++i) {
        const uint32_t pk = reg_get(c, i);
        const int px = (int)(pk & 0xffffu), py = (int)(pk >> 16);
        if (dist_sq(xc, yc, (double)px, (double)py) > radSq) {
            used_clr(c, find_e(c, px, py));
            const uint32_t last = reg_get(c, reg_size - 1);
            reg_set(c, i, last);
            reg_set(c, reg_size - 1, pk);
            mem_fence();
            --reg_size;
            --i;
        }
    }
#endif
}

// First half of refine(): the region's points are released (USED cleared) and the tolerance tau for the second
// growth is twice the standard deviation of the angles near the region's first point.
LFG_DEV double refine_tau(const Ctx& c, int reg_size, const Rect& rec, int& x0, int& y0, int& e0)
{
    const int lane = lane_id();
    const uint32_t p0 = reg_get(c, 0);
    x0 = (int)(p0 & 0xffffu); y0 = (int)(p0 >> 16);
    const double xc = (double)x0, yc = (double)y0;
    e0 = find_e(c, x0, y0);
    const double ang_c = angle_of(c.deg[e0]);
    double sum = 0, s_sum = 0;
    int n = 0;
#ifndef LF_HOST_SIM
    // Every lane tests its own point (distance from the first point, a square root) and forms its own angle difference and square; the
    // two ordered sums then take the selected points' terms in list order through ordered_sums3 -- a point outside the radius
    // contributes -0.0, which leaves a sum as it is.  (Before: one point at a time on wave-uniform values, square root included.)
    double acc = 0.0;
    for (int base = 0; base < reg_size; base += LFG_NL) {
        const int i = base + lane;
        const bool v = i < reg_size;
        const uint32_t pk = v ? reg_get(c, i) : 0u;
        const int px = (int)(pk & 0xffffu), py = (int)(pk >> 16);
        const int ei = v ? find_e(c, px, py) : 0;
        const float af = v ? c.deg[ei] : NOTDEF_F;
        const int cnt = reg_size - base < LFG_NL ? reg_size - base : LFG_NL;
        if (v) used_and(c, ei);                        // every lane releases its own point: the order of the releases means nothing
        const bool sel = v && dist_(xc, yc, (double)px, (double)py) < rec.width;
        double ang_d = -0.0, ang_d2 = -0.0;
        if (sel) {
            double diff = angle_of(af) - ang_c;        // angle_diff_signed, per lane
            while (diff <= -PI_) diff += M_2__PI_;
            while (diff > PI_) diff -= M_2__PI_;
            ang_d = diff; ang_d2 = diff * diff;
        }
        ordered_sums3(acc, ang_d, ang_d2, -0.0, cnt);
        n += __popcll(lfg_ballot(sel));
    }
    sum = rl_d(acc, 0); s_sum = rl_d(acc, 1);
#else
    for (int base = 0; base < reg_size; base += LFG_NL) {
        const int i = base + lane;
        const bool v = i < reg_size;
        const uint32_t pk = v ? reg_get(c, i) : 0u;
        const int px = (int)(pk & 0xffffu), py = (int)(pk >> 16);
        const int ei = v ? find_e(c, px, py) : 0;
        const float af = v ? c.deg[ei] : NOTDEF_F;
        const int cnt = reg_size - base < LFG_NL ? reg_size - base : LFG_NL;
        for (int j = 0; j < cnt; ++j) {
            const uint32_t q = (uint32_t)rl_i((int)pk, j);
            const int qx = (int)(q & 0xffffu), qy = (int)(q >> 16);
#ifdef LF_HOST_SIM
            used_clr(c, rl_i(ei, j));
#endif
            if (dist_(xc, yc, (double)qx, (double)qy) < rec.width) {
                const double angle = angle_of(rl_f(af, j));
                const double ang_d = angle_diff_signed(angle, ang_c);
                sum += ang_d;
                s_sum += ang_d * ang_d;
                ++n;
            }
        }
    }
#endif
    const double mean_angle = sum / (double)n;
    return 2.0 * dm::dsqrt((s_sum - 2.0 * mean_angle * sum) / (double)n + mean_angle * mean_angle);
}

// ------------------------------------------------------------------ NFA
LFG_DEV double log_gamma_lanczos(double x)
{
    const double q[7] = { 75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705,
                          1168.92649479, 83.8676043424, 2.50662827511 };
    double a = (x + 0.5) * dm::dlog(x + 5.5) - (x + 5.5);
    double b = 0;
#pragma unroll
    for (int n = 0; n < 7; ++n) {
        a -= dm::dlog(x + (double)n);
        b += q[n] * dm::dpow(x, (double)n);
    }
    return a + dm::dlog(b);
}
LFG_DEV double log_gamma_windschitl(double x)
{
    return 0.918938533204673 + (x - 0.5) * dm::dlog(x) - x
         + 0.5 * x * dm::dlog(x * dm::dsinh_small(1 / x) + 1 / (810.0 * dm::dpow(x, 6.0)));
}
LFG_DEV double log_gamma(double x) { return x > 15.0 ? log_gamma_windschitl(x) : log_gamma_lanczos(x); }

// The NFA arithmetic (log-gamma, binomial tail: a few hundred f64 operations through the deterministic routines) is
// reached from six places once everything is inlined; kept OUT of line it exists once, and the kernel's hot loops
// stay inside the instruction cache (the kernel was 116 KB of code against a 64 KB cache shared by two CUs).
#ifndef LF_HOST_SIM
__device__ __noinline__
#else
static
#endif
double nfa_eval(int n, int k, double p, double LOG_NT)
{
    if (n == 0 || k == 0) return -LOG_NT;
    if (n == k) return -LOG_NT - (double)n * dm::dlog10(p);
    const double p_term = p / (1 - p);
    const double log1term = ((double)n + 1) - log_gamma((double)k + 1) - log_gamma((double)(n - k) + 1)
                          + (double)k * dm::dlog(p) + (double)(n - k) * dm::dlog(1.0 - p);
    double term = dm::dexp(log1term);
    if (double_equal(term, 0)) {
        if (k > n * p) return -log1term / 2.30258509299404568402 - LOG_NT;
        else return -LOG_NT;
    }
    double bin_tail = term;
    const double tolerance = 0.1;
    for (int i = k + 1; i <= n; ++i) {
        const double bin_term = (double)(n - i + 1) / (double)i;
        const double mult_term = bin_term * p_term;
        term *= mult_term;
        bin_tail += term;
        if (bin_term < 1) {
            const double err = term * ((1 - dm::dpow(mult_term, (double)(n - i + 1))) / (1 - mult_term) - 1);
            if (err < tolerance * dabs(-dm::dlog10(bin_tail) - LOG_NT) * bin_tail) break;
        }
    }
    return -dm::dlog10(bin_tail) - LOG_NT;
}
LFG_DEV double nfa(const Ctx& c, int n, int k, double p) { return nfa_eval(n, k, p, c.log_nt); }

// Row geometry of one rectangle: everything rect_nfa's scan-line walk needs, all integers.
struct RowGeom { int min_x, flstep, slstep, frstep, srstep, ly, ry, y_start, y_end; };

LFG_DEV RowGeom edge_setup(const Rect& rec, int H)
{
    const double half_width = rec.width / 2.0;
    const double dyhw = rec.dy * half_width;
    const double dxhw = rec.dx * half_width;
    int ex[4], ey[4];
    ex[0] = (int)(rec.x1 - dyhw); ey[0] = (int)(rec.y1 + dxhw);
    ex[1] = (int)(rec.x2 - dyhw); ey[1] = (int)(rec.y2 + dxhw);
    ex[2] = (int)(rec.x2 + dyhw); ey[2] = (int)(rec.y2 - dxhw);
    ex[3] = (int)(rec.x1 + dyhw); ey[3] = (int)(rec.y1 - dxhw);
    // sort 4 corners by (x, y) with a fixed compare-exchange network (same result as the
    // oracle's insertion sort: equal keys are identical points)
#define LFG_CX(a, b)                                                                   \
    {                                                                                  \
        bool sw = (ex[b] < ex[a]) || (ex[b] == ex[a] && ey[b] < ey[a]);                \
        int tx = sw ? ex[b] : ex[a], ty = sw ? ey[b] : ey[a];                          \
        int ux = sw ? ex[a] : ex[b], uy = sw ? ey[a] : ey[b];                          \
        ex[a] = tx; ey[a] = ty; ex[b] = ux; ey[b] = uy;                                \
    }
    LFG_CX(0, 1) LFG_CX(2, 3) LFG_CX(0, 2) LFG_CX(1, 3) LFG_CX(1, 2)
#undef LFG_CX
    int imin = 0;
    int min_x = ex[0], min_yv = ey[0], max_yv = ey[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        if (min_yv > ey[i]) { min_yv = ey[i]; min_x = ex[i]; imin = i; }   // first minimum in sorted order
        if (max_yv < ey[i]) max_yv = ey[i];
    }
    bool taken[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) taken[i] = (i == imin);
    int lx = 0, ly = 0, rx = 0, ry = 0, tx_ = 0;
    {
        int sel = -1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (!taken[i]) { if (sel < 0) { sel = i; lx = ex[i]; ly = ey[i]; } else if (lx > ex[i]) { sel = i; lx = ex[i]; ly = ey[i]; } }
#pragma unroll
        for (int i = 0; i < 4; ++i) if (i == sel) taken[i] = true;
        sel = -1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (!taken[i]) { if (sel < 0) { sel = i; rx = ex[i]; ry = ey[i]; } else if (rx < ex[i]) { sel = i; rx = ex[i]; ry = ey[i]; } }
#pragma unroll
        for (int i = 0; i < 4; ++i) if (i == sel) taken[i] = true;
        sel = -1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (!taken[i]) { if (sel < 0) { sel = i; tx_ = ex[i]; } else if (tx_ > ex[i]) { sel = i; tx_ = ex[i]; } }
    }
    RowGeom g;
    // integer edge steps and the y-vs-x comparisons of OpenCV 3.x, kept as in the oracle
    g.min_x = min_x;
    g.flstep = (min_yv != ly) ? (min_x - lx) / (min_yv - ly) : 0;
    g.slstep = (ly != tx_) ? (lx - tx_) / (ly - tx_) : 0;
    g.frstep = (min_yv != ry) ? (min_x - rx) / (min_yv - ry) : 0;
    g.srstep = (ry != tx_) ? (rx - tx_) / (ry - tx_) : 0;
    g.ly = ly; g.ry = ry;
    // The oracle walks rows sequentially, skipping (without stepping) rows outside the image.
    g.y_start = min_yv > 0 ? min_yv : 0;
    g.y_end = max_yv < H - 1 ? max_yv : H - 1;
    return g;
}

// The rectangle is the same in every lane: tell the compiler, so the nine fields live in scalar registers.
LFG_DEV RowGeom uniform_geom(const RowGeom& g)
{
#ifndef LF_HOST_SIM
    RowGeom u;
    u.min_x = __builtin_amdgcn_readfirstlane(g.min_x); u.flstep = __builtin_amdgcn_readfirstlane(g.flstep);
    u.slstep = __builtin_amdgcn_readfirstlane(g.slstep); u.frstep = __builtin_amdgcn_readfirstlane(g.frstep);
    u.srstep = __builtin_amdgcn_readfirstlane(g.srstep); u.ly = __builtin_amdgcn_readfirstlane(g.ly);
    u.ry = __builtin_amdgcn_readfirstlane(g.ry); u.y_start = __builtin_amdgcn_readfirstlane(g.y_start);
    u.y_end = __builtin_amdgcn_readfirstlane(g.y_end);
    return u;
#else
    return g;
#endif
}

// Pixel span [xa, xb] of row y (y_start <= y <= y_end), clipped to the image.  All quantities are
// integers, so the oracle's step-by-step edge walk has this closed form.
LFG_DEV void row_span(const RowGeom& g, int y, int W, int& xa, int& xb)
{
    const int steps = y - g.y_start;
    int nf = (y < g.ly ? y : g.ly) - g.y_start; nf = nf < 0 ? 0 : (nf > steps ? steps : nf);
    int nr = (y < g.ry ? y : g.ry) - g.y_start; nr = nr < 0 ? 0 : (nr > steps ? steps : nr);
    const int xl = g.min_x + g.flstep * nf + g.slstep * (steps - nf);
    const int xr = g.min_x + g.frstep * nr + g.srstep * (steps - nr);
    xa = xl < 0 ? 0 : xl;
    xb = xr > W - 1 ? W - 1 : xr;
}

// first entry of row y whose x is >= xa (binary search in the row's sorted x list)
LFG_DEV int row_lower_bound(const Ctx& c, int y, int xa)
{
    if (bm_mode(c)) return rank_before(c, y * c.W + xa);
    int lo = c.rows[y], hi = c.rows[y + 1];
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (xs_get(c, mid) < xa) lo = mid + 1; else hi = mid; }
    return lo;
}

// folded angle distance used by isAligned (entries always have a defined angle)
LFG_DEV double angle_dist(float deg, double theta)
{
    double n_theta = theta - (double)deg * DEG2RAD;
    if (n_theta < 0) n_theta = -n_theta;
    if (n_theta > M_3_2_PI_) {
        n_theta -= M_2__PI_;
        if (n_theta < 0) n_theta = -n_theta;
    }
    return n_theta;
}

// Aligned-point count + NFA of one rectangle: the wave's lanes take rows y_start+lane, +64, ...; a
// row's aligned pixels are found in its sorted defined-pixel list (LDS) instead of the 96 %
// undefined angle plane.  Integer counts: order free.
LFG_DEV double rect_nfa(const Ctx& c, const Rect& rec)
{
    LFG_T0
    const RowGeom g = edge_setup(rec, c.H);
    int total_pts = 0, alg_pts = 0;
    for (int y = g.y_start + lane_id(); y <= g.y_end; y += LFG_NL) {
        int xa, xb;
        row_span(g, y, c.W, xa, xb);
        if (xb < xa) continue;
        total_pts += xb - xa + 1;
        const int lo = row_lower_bound(c, y, xa);
        if (bm_mode(c)) {
            // the row's defined pixels in [xa, xb] are the entries lo .. hi - 1: no coordinates needed
            const int hi = rank_before(c, y * c.W + xb + 1);
            for (int k0 = lo; k0 < hi; k0 += 4) {
                float dv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) dv[u] = c.deg[k0 + u < hi ? k0 + u : hi - 1];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (k0 + u < hi && angle_dist(dv[u], rec.theta) <= rec.prec) ++alg_pts;
            }
            continue;
        }
        const int e1 = c.rows[y + 1];
        for (int k0 = lo; k0 < e1; k0 += 4) {
            int xv[4];
            float dv[4];
            bool stop = false;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = k0 + u < e1 ? k0 + u : e1 - 1;
                xv[u] = xs_get(c, idx);
                dv[u] = c.deg[idx];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (k0 + u >= e1 || xv[u] > xb) { stop = true; break; }
                if (angle_dist(dv[u], rec.theta) <= rec.prec) ++alg_pts;
            }
            if (stop) break;
        }
    }
    total_pts = wave_sum_i(total_pts);
    alg_pts = wave_sum_i(alg_pts);
    LFG_T1(c, 4)
#if defined(LFG_STAMPS) && !defined(LF_HOST_SIM)
    c.stamps[7] += (1ull << 40) + (unsigned long long)total_pts;
#endif
    const double r_ = nfa(c, total_pts, alg_pts, rec.p);
    LFG_T1(c, 5)
    return r_;
}

// The five candidate rectangles of one rect_improve stage at once.  They share theta and differ by
// half a pixel of width / offset or by the precision only, so their rows and x spans nearly coincide:
// ONE pass over the union of the spans (all 64 lanes on rows, one angle fetch per entry) feeds five
// pairs of counters; the five NFA values -- log-gamma, binomial tail -- are then computed SIMD, one
// candidate per 8-lane group.
// Candidate n of a rect_improve stage: the stage's step applied n+1 times to the stage's starting
// rectangle (the reference's inner loops are cumulative).  Returns false when the reference would
// skip the candidate (width cannot shrink further); the rectangle is then left as it was.
LFG_DEV bool improve_step(Rect& r, int stage)
{
    const double delta = 0.5;
    const double delta_2 = delta / 2.0;
    if (stage == 0) {
        r.p /= 2;
        r.prec = r.p * PI_;
        return true;
    }
    if (!((r.width - delta) >= 0.5)) return false;     // OpenCV keeps this guard on the last stage too
    if (stage == 4) {
        r.p /= 2;
        r.prec = r.p * PI_;
        return true;
    }
    if (stage == 2) {
        r.x1 += -r.dy * delta_2;
        r.y1 += r.dx * delta_2;
        r.x2 += -r.dy * delta_2;
        r.y2 += r.dx * delta_2;
    } else if (stage == 3) {
        r.x1 -= -r.dy * delta_2;
        r.y1 -= r.dx * delta_2;
        r.x2 -= -r.dy * delta_2;
        r.y2 -= r.dx * delta_2;
    }
    r.width -= delta;
    return true;
}
LFG_DEV bool improve_candidate(const Rect& base, int stage, int n, Rect& r)
{
    r = base;
    bool ok = true;
    for (int j = 0; j <= n; ++j) ok = improve_step(r, stage);
    return ok;
}

LFG_DEV void rect_nfa5(const Ctx& c, const Rect& base, int stage, bool* valid, double* v)
{
    LFG_T0
    RowGeom G[5];
    double prec[5], pp[5];
    int y_lo = c.H, y_hi = -1;
    {
        Rect r = base;
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            if (ok) ok = improve_step(r, stage);        // once a step is refused the rectangle stays put
            valid[k] = ok;
            G[k] = uniform_geom(edge_setup(r, c.H));
            prec[k] = r.prec; pp[k] = r.p;
            if (ok) {
                y_lo = G[k].y_start < y_lo ? G[k].y_start : y_lo;
                y_hi = G[k].y_end > y_hi ? G[k].y_end : y_hi;
            }
        }
    }
    const double theta = base.theta;                 // rect_improve never changes theta
    int tot[5] = { 0, 0, 0, 0, 0 }, alg[5] = { 0, 0, 0, 0, 0 };
    for (int y = y_lo + lane_id(); y <= y_hi; y += LFG_NL) {
        int xa[5], xb[5];
        int ua = c.W, ub = -1;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            xa[k] = 1; xb[k] = 0;
            if (valid[k] && y >= G[k].y_start && y <= G[k].y_end) {
                row_span(G[k], y, c.W, xa[k], xb[k]);
                if (xb[k] >= xa[k]) {
                    tot[k] += xb[k] - xa[k] + 1;
                    ua = xa[k] < ua ? xa[k] : ua;
                    ub = xb[k] > ub ? xb[k] : ub;
                }
            }
        }
        if (ub < ua) continue;
        const int lo = row_lower_bound(c, y, ua);
        if (bm_mode(c)) {
            // walk the set bits of the row between ua and ub: the bit's place is the pixel's x, the running rank its entry
            const int p0 = y * c.W + ua, p1 = y * c.W + ub;
            int idx = lo;
            for (int w = p0 >> 6; w <= (p1 >> 6); ++w) {
                unsigned long long bw = c.bits64[w];
                if (w == (p0 >> 6)) bw &= ~0ull << (p0 & 63);
                if (w == (p1 >> 6) && (p1 & 63) != 63) bw &= (2ull << (p1 & 63)) - 1ull;
                while (bw) {
                    int xv[4], cnt = 0;
                    float dv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        xv[u] = 0; dv[u] = 0.f;
                        if (bw) {
                            xv[u] = ((w << 6) | __builtin_ctzll(bw)) - y * c.W;
                            bw &= bw - 1ull;
                            dv[u] = c.deg[idx + u];
                            cnt = u + 1;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (u >= cnt) break;
                        const double nt = angle_dist(dv[u], theta);
#pragma unroll
                        for (int k = 0; k < 5; ++k)
                            if (xv[u] >= xa[k] && xv[u] <= xb[k] && nt <= prec[k]) ++alg[k];
                    }
                    idx += cnt;
                }
            }
            continue;
        }
        const int e1 = c.rows[y + 1];
        for (int k0 = lo; k0 < e1; k0 += 4) {
            int xv[4];
            float dv[4];
            bool stop = false;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = k0 + u < e1 ? k0 + u : e1 - 1;
                xv[u] = xs_get(c, idx);
                dv[u] = c.deg[idx];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (k0 + u >= e1 || xv[u] > ub) { stop = true; break; }
                const double nt = angle_dist(dv[u], theta);
#pragma unroll
                for (int k = 0; k < 5; ++k)
                    if (xv[u] >= xa[k] && xv[u] <= xb[k] && nt <= prec[k]) ++alg[k];
            }
            if (stop) break;
        }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) { tot[k] = wave_sum_i(tot[k]); alg[k] = wave_sum_i(alg[k]); }
    LFG_T1(c, 4)
#if defined(LFG_STAMPS) && !defined(LF_HOST_SIM)
    c.stamps[7] += (1ull << 40) + (unsigned long long)tot[0];
#endif
#ifndef LF_HOST_SIM
    const int g = lane_id() >> 3;
    int n_ = tot[0], k_ = alg[0];
    double p_ = pp[0];
    bool on = valid[0];
#pragma unroll
    for (int k = 1; k < 5; ++k)
        if (g == k) { n_ = tot[k]; k_ = alg[k]; p_ = pp[k]; on = valid[k]; }
    if (g > 4) on = false;
    const double r = on ? nfa(c, n_, k_, p_) : -1e300;
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] = rl_d(r, 8 * k);
#else
    for (int k = 0; k < 5; ++k) v[k] = valid[k] ? nfa(c, tot[k], alg[k], pp[k]) : -1e300;
#endif
    LFG_T1(c, 5)
}

// rect_improve: the reference tries 5 x 5 variants one after another, but a stage's variants
// depend only on the stage's starting rectangle, never on each other's NFA -- so each stage is
// evaluated as one 5-wide rect_nfa5 and then folded in the reference's order.
LFG_DEV double rect_improve(const Ctx& c, Rect& rec)
{
    const double LOG_EPS = c.log_eps;
    double log_nfa = rect_nfa(c, rec);
    if (log_nfa > LOG_EPS) return log_nfa;
    bool valid[5];
    double v[5];
    for (int stage = 0; stage < 5; ++stage) {
        rect_nfa5(c, rec, stage, valid, v);
        int best = -1;
#pragma unroll
        for (int n = 0; n < 5; ++n)
            if (valid[n] && v[n] > log_nfa) { log_nfa = v[n]; best = n; }
        if (best >= 0) {
            Rect r;
            improve_candidate(rec, stage, best, r);
            rec = r;
        }
        if (stage < 4 && log_nfa > LOG_EPS) return log_nfa;
    }
    return log_nfa;
}

// ------------------------------------------------------------------ main loop (flsd)
// order: sorted seed items ((1023-bin) << 20 | compact entry e); returns the number of lines this call found.
// Lines go to slot atomicAdd(*c.line_count) while that is below cap, with their seed's position in `order` as tag.
// rect_improve + emission of one finished region (in place, or by a helping wave from the ring)
// IMPROVE = false: the caller knows that refine < 2 (no rect_improve: nothing but the emission is instantiated)
template <bool IMPROVE>
LFG_DEV bool evaluate_region(const Ctx& c, Rect& rec, int tag, float* lines, int cap, int n_lines)
{
    if (IMPROVE && c.refine >= 2) {
        const double log_nfa = rect_improve(c, rec);
        if (log_nfa <= c.log_eps) return false;
    }
    rec.x1 += 0.5; rec.y1 += 0.5; rec.x2 += 0.5; rec.y2 += 0.5;
    if (c.scale != 1) {
        rec.x1 /= c.scale; rec.y1 /= c.scale; rec.x2 /= c.scale; rec.y2 /= c.scale;
    }
#ifndef LF_HOST_SIM
    (void)n_lines;
    // lane 0 takes the slot, EVERY lane stores the (same) line: a function that ends in lane-0-only code is not safe at the end
    // of a loop body (see evaluate_pending)
    int slot = 0;
    if (lane_id() == 0) slot = atomicAdd(c.line_count, 1);     // any order: the tags restore the sequential one
    slot = __builtin_amdgcn_readfirstlane(slot);
    if (slot < cap) {
        *reinterpret_cast<float4*>(lines + 4 * slot) = make_float4((float)rec.x1, (float)rec.y1, (float)rec.x2, (float)rec.y2);
        c.tags[slot] = tag;
    }
#else
    {
        const int slot = c.line_count ? (*c.line_count)++ : n_lines;
        if (slot < cap) {
            lines[4 * slot + 0] = (float)rec.x1;
            lines[4 * slot + 1] = (float)rec.y1;
            lines[4 * slot + 2] = (float)rec.x2;
            lines[4 * slot + 3] = (float)rec.y2;
            if (c.tags) c.tags[slot] = tag;
        }
    }
#endif
    return true;
}

#ifndef LF_HOST_SIM
// k_lsd_eval's form of evaluate_region: rect_improve and the final coordinates of one pending region, WITHOUT any
// lane-dependent control flow -- every lane returns the same answer and the caller stores it from all lanes.  (A loop
// whose body ends in `if (lane == 0) { emit }` is not safe: the compiler sent lanes 1..63 straight back to the loop
// header with their own, never-assigned ticket while lane 0 was still emitting, and the wave never came out again.)
LFG_DEV bool evaluate_pending(const Ctx& c, Rect& rec, float4& line)
{
    const double log_nfa = rect_improve(c, rec);
    if (log_nfa <= c.log_eps) return false;
    rec.x1 += 0.5; rec.y1 += 0.5; rec.x2 += 0.5; rec.y2 += 0.5;
    if (c.scale != 1) {
        rec.x1 /= c.scale; rec.y1 /= c.scale; rec.x2 /= c.scale; rec.y2 /= c.scale;
    }
    line = make_float4((float)rec.x1, (float)rec.y1, (float)rec.x2, (float)rec.y2);
    return true;
}

// LFG_EVAL_KERNEL: one more entry of the problem's pending list (lane 0 stores; the rectangle is wave-uniform)
LFG_DEV void eval_append(const Ctx& c, const Rect& rec, int tag)
{
    int slot = 0;
    if (lane_id() == 0) slot = atomicAdd(c.pend_n, 1);
    slot = __builtin_amdgcn_readfirstlane(slot);
    if (slot < c.pend_cap) {                                   // every lane stores the same entry (no lane-0-only tail)
        double2* d = reinterpret_cast<double2*>(c.pend_rec + (size_t)slot * 12);
        d[0] = make_double2(rec.x1, rec.y1); d[1] = make_double2(rec.x2, rec.y2); d[2] = make_double2(rec.width, rec.x);
        d[3] = make_double2(rec.y, rec.theta); d[4] = make_double2(rec.dx, rec.dy); d[5] = make_double2(rec.prec, rec.p);
        c.pend_tag[slot] = tag;
    }
}

LFG_DEV bool eval_push(const Ctx& c, const Rect& rec, int tag)
{
    EvalQueue* q = c.q;
    int slot = -1;
    if (lane_id() == 0) {
        int t = __hip_atomic_load(&q->tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        for (;;) {
            const int h = __hip_atomic_load(&q->head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (t - h >= LFG_QN) break;                                                                    // full
            if (__hip_atomic_load(&q->seq[t % LFG_QN], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) break;   // still being copied out
            const int old = atomicCAS(&q->tail, t, t + 1);
            if (old == t) { slot = t; break; }
            t = old;
        }
        if (slot >= 0) {
            const int s = slot % LFG_QN;
            double* d = q->rec[s];
            d[0] = rec.x1; d[1] = rec.y1; d[2] = rec.x2; d[3] = rec.y2; d[4] = rec.width; d[5] = rec.x; d[6] = rec.y;
            d[7] = rec.theta; d[8] = rec.dx; d[9] = rec.dy; d[10] = rec.prec; d[11] = rec.p;
            q->tag[s] = tag;
            __hip_atomic_store(&q->seq[s], slot + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    return __builtin_amdgcn_readfirstlane(slot) >= 0;
}

LFG_DEV bool eval_pop(const Ctx& c, Rect& rec, int& tag)
{
    EvalQueue* q = c.q;
    int t = -1;
    if (lane_id() == 0) {
        int h = __hip_atomic_load(&q->head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        for (;;) {
            if (h >= __hip_atomic_load(&q->tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;   // empty
            const int old = atomicCAS(&q->head, h, h + 1);
            if (old == h) { t = h; break; }
            h = old;
        }
        if (t >= 0) {
            // the producer of ticket t reserved it before it wrote the item: a few stores away at most (bounded spin: a
            // ring gone wrong must not hang the GPU; the parity tests would show the lost region)
            int spins = 0;
            while (__hip_atomic_load(&q->seq[t % LFG_QN], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != t + 1 && ++spins < (1 << 20))
                __builtin_amdgcn_s_sleep(1);
        }
    }
    t = __builtin_amdgcn_readfirstlane(t);
    if (t < 0) return false;
    const int s = t % LFG_QN;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const double* d = q->rec[s];
    rec.x1 = d[0]; rec.y1 = d[1]; rec.x2 = d[2]; rec.y2 = d[3]; rec.width = d[4]; rec.x = d[5]; rec.y = d[6];
    rec.theta = d[7]; rec.dx = d[8]; rec.dy = d[9]; rec.prec = d[10]; rec.p = d[11];
    tag = q->tag[s];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane_id() == 0) __hip_atomic_store(&q->seq[s], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    return true;
}
#endif

LFG_DEV int detect(const Ctx& c, const uint32_t* order, int n_order, float* lines, int cap)
{
    int n_lines = 0;
    LFG_T0
#ifndef LF_HOST_SIM
    // Seeds are fetched 64 at a time (one coalesced load + their coordinates).  Every lane tests the
    // USED bit of its own seed; the next seed the reference would start from is the lowest lane past
    // the cursor whose pixel is free NOW (a region can free pixels again in refine(), so the test is
    // repeated after every region) -- one vector test + ballot per region instead of a scalar
    // test per seed.
    for (int base = 0; base < n_order; base += LFG_NL) {
        LFG_T1(c, 0)
        bool sv = base + lane_id() < n_order;
        const uint32_t seed_items = sv ? order[base + lane_id()] : 0u;
        const int my_e = (int)(seed_items & 0xfffffu);
        const uint32_t seed_xy = sv ? c.gxy[my_e] : 0u;               // beside the label, not behind it: one round trip per chunk less
        const float seed_dg = sv ? c.deg[my_e] : 0.f;                  // ... and the seed's angle (region_grow: pre_deg)
        if (sv && c.label) sv = (int)c.label[my_e] == c.root;          // seeds of other components are not ours
        const unsigned long long svm = lfg_ballot(sv);                 // our seeds of this chunk, as a lane mask (see the accept loop)
        if (svm == 0ull) continue;
        LFG_T1(c, 8)
        unsigned long long pending = ~0ull;
      for (;;) {
        mem_fence();
        const unsigned long long fr = svm & ~lfg_ballot(used_get(c, my_e)) & pending;      // (lanes without a seed read entry 0: masked)
        if (fr == 0ull) break;
        const int sl = __builtin_ctzll(fr);
        pending = sl >= 63 ? 0ull : (~0ull << (sl + 1));
        const int se = rl_i(my_e, sl);
        const uint32_t sxy = (uint32_t)rl_i((int)seed_xy, sl);
        const float sdg = rl_f(seed_dg, sl);
        const int tag = base + sl;
        LFG_T1(c, 0)
#else
    for (int i = 0; i < n_order; ++i) {
      {
        const int se = (int)(order[i] & 0xfffffu);
        if (c.label && (int)c.label[se] != c.root) continue;
        if (used_get(c, se)) continue;
        const uint32_t sxy = c.gxy[se];
        const int tag = i;
#endif
        // region_grow -> region2rect -> [refine: second growth with tau -> region2rect] -> [radius reduction steps, each
        // followed by region2rect], as one loop (see "refine" above).  phase 0: first growth, 1: grown again with tau,
        // 2: reducing the radius.
        int reg_size = 0;
        double reg_angle = 0;
        Rect rec;
        int phase = 0, gx = (int)(sxy & 0xffffu), gy = (int)(sxy >> 16), ge = se;
        double grow_prec = c.prec, radSq = 0, xc = 0, yc = 0;
        bool rejected = false;
        for (;;) {
            if (phase < 2) {
#ifndef LF_HOST_SIM
                region_grow(c, gx, gy, ge, reg_size, reg_angle, grow_prec, phase == 0 ? sdg : -4096.f);
#else
                region_grow(c, gx, gy, ge, reg_size, reg_angle, grow_prec);
#endif
                if (phase == 0) { LFG_T1(c, 1) LFG_CNT(c, 9, 1) LFG_CNT(c, 10, reg_size) } else { LFG_T1(c, 22) }
                reg_size = uni_i(reg_size);
                if (reg_size < (phase == 0 ? c.min_reg_size : 2)) { rejected = true; break; }
            }
            region2rect(c, reg_size, reg_angle, c.prec, c.p, rec);
            if (phase == 0) { LFG_T1(c, 2) } else { LFG_T1(c, 17) }
            if (c.refine <= 0) break;
            const double density = (double)reg_size / (dist_(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
            if (uni(density >= c.density_th)) break;
            if (phase == 0) {
                grow_prec = refine_tau(c, reg_size, rec, gx, gy, ge);
                LFG_T1(c, 21)
                phase = 1;
                continue;
            }
            if (phase == 1) {
                const uint32_t p0 = reg_get(c, 0);
                xc = (double)(int)(p0 & 0xffffu); yc = (double)(int)(p0 >> 16);
                const double radSq1 = dist_sq(xc, yc, rec.x1, rec.y1);
                const double radSq2 = dist_sq(xc, yc, rec.x2, rec.y2);
                radSq = radSq1 > radSq2 ? radSq1 : radSq2;
                phase = 2;
            }
            reduce_radius_step(c, reg_size, xc, yc, radSq);
            reg_size = uni_i(reg_size);
            LFG_T1(c, 23) LFG_CNT(c, 20, 0)
            if (reg_size < 2) { rejected = true; break; }
        }
        LFG_T1(c, 3)
        if (rejected) continue;
#if !defined(LF_HOST_SIM) && LFG_EVAL_KERNEL
        // rect_improve and the emission happen in k_lsd_eval, from the problem's pending list
        if (c.refine >= 2) { eval_append(c, rec, tag); continue; }
        if (!evaluate_region<false>(c, rec, tag, lines, cap, n_lines)) { LFG_T1(c, 6) continue; }
#elif !defined(LF_HOST_SIM) && LFG_EVAL_QUEUE
        // rect_improve lives in ONE place, the helper loop of k_lsd_grow.hip (a second inlined copy here made the kernel
        // 71 KB of code against a 64 KB instruction cache): regions that need it always go through the ring.  The
        // workgroup's last wave never grows, so a full ring (sixteen regions waiting: the helper would have to fall
        // 16 x 60 k cycles behind) only ever means a short wait.
        if (c.refine >= 2) {
            while (!eval_push(c, rec, tag)) __builtin_amdgcn_s_sleep(16);
            LFG_T1(c, 31)
            continue;
        }
        if (!evaluate_region<false>(c, rec, tag, lines, cap, n_lines)) { LFG_T1(c, 6) continue; }
#else
        if (!evaluate_region<true>(c, rec, tag, lines, cap, n_lines)) { LFG_T1(c, 6) continue; }
#endif
        ++n_lines;
        LFG_T1(c, 6)
      }
    }
    LFG_T1(c, 0)
    return n_lines;
}

}  // namespace grow
}  // namespace lf
