// K_lsd_grow: launches lsd_grow.h -- one 64-lane workgroup per (frame, colour) LSD problem.
// See lsd_grow.h for the algorithm, the reference citation and the lane-cooperation scheme.
#include "common.h"
#include "lsd_grow.h"
#include "lsd_bitplane.h"

namespace lf {

// Occupancy knobs.  The kernel is latency bound (one wave per component, long dependent chains) and the pipelined
// rate follows the number of growing waves the chip can hold.  With rect_improve in the evaluating wave's loop only
// (below) the growing code needs far fewer registers than it did: at 96 VGPRs (five waves per SIMD) 19 values spill,
// at 80 (SIX waves per SIMD = six four-wave problems per CU) a few more, and the pipelined bench gains 2.3 %
// (same-call A/B: 141.8 k -> 145.0 k frames/s; kernel alone 2.84 -> 2.89 ms; 72 / 64 VGPRs: 146 k / 145 k on a box where
// 80 gave 146.6 k).  History: 128 natural -> 96 at three waves per problem was +2.2 %; capping the kernel at 12 waves per
// CU instead of 16 cost 12 %.  13 KB of LDS per problem stays above the wave limit.
// Round 4, after the bit plane (25 KB of LDS per problem: six problems per CU by LDS) and the work on the growing wave's
// instruction stream: FIVE waves per SIMD = 96 VGPRs (38 spilled values instead of 129) = five problems per CU measures better than six
// (same-call A/B, two pairs: lane frames 151.9 / 152.0 k -> 153.3 / 154.4 k frames/s, clutter 38.0 / 38.4 k -> 39.8 / 40.0 k, camera
// frames equal; four waves = 128 VGPRs, no spills: 149.3 k / 36.5 k / 77 k) -- and leaves 35 KB of LDS per CU to the other kernels.
#ifndef LFG_WAVES
#define LFG_WAVES 5
#endif
#ifndef LFG_LDS_KB
#define LFG_LDS_KB 13
#endif
#ifndef LFG_REG_LDS
#define LFG_REG_LDS 256        // region-list entries per wave kept in LDS (the rest of a long region goes to the wave's scratch slice):
#endif                         // 256 instead of 512 = 3 KB less per problem, +1 % frames/s (other kernels' workgroups find LDS sooner)

// Deferred evaluation (lsd_grow.h EvalQueue): FOUR waves per problem, three grow components and the last one only
// evaluates the finished regions the growers push into the ring (it stays for as long as anyone grows, polling between
// s_sleeps; a grower that has finished drains what is there and leaves, its wave slot goes to another problem).  With a
// wave that is always there to evaluate, rect_improve is instantiated ONCE, in the helper loop below: a grower never
// evaluates in place (it waits the few cycles a full ring would take to move), and the kernel is 44 KB of code instead
// of the 71 KB it was with a second inlined copy in detect() -- the instruction cache is 64 KB for two CUs.
// Measured in the same call, configs[1], against "first idle wave of three helps, in-place evaluation when the ring is
// full" (r02_l .. r02_o): kernel alone 2.95 -> 2.82 ms, six batches in flight 139.3k -> 141.4k frames/s.  Two growers
// + helper: 3.39 ms, 131.4k (problems with three or more components lose a grower).  r02_l's history: ring + first
// idle wave of three vs. no ring: longest problem 7.96 -> 6.92 Mcycles, 127.5k -> 130.1k frames/s.
//
// LFG_EVAL_KERNEL=1 (lsd_grow.h; measured in round 3, not the default): the evaluation in a kernel of its own, k_lsd_eval
// below -- THREE waves per problem, all growing; finished regions go to the problem's pending list in HBM.
#ifndef LFG_GROW_WAVES
#define LFG_GROW_WAVES (LFG_EVAL_KERNEL ? 3 : 4)
#endif
#ifndef LFG_EVAL_WAVES
#define LFG_EVAL_WAVES 4        // waves of an evaluation workgroup: pending regions are handed out one at a time
#endif
static_assert(LFG_GROW_WAVES >= 2 || !LFG_EVAL_QUEUE, "with the evaluation ring the last wave of a workgroup never grows");
constexpr int GROW_WAVES = LFG_GROW_WAVES;   // waves per problem: components of the defined-pixel graph are handed out among the growing ones
constexpr int GROW_LISTS = LFG_EVAL_QUEUE ? GROW_WAVES - 1 : GROW_WAVES;    // region lists (LDS + scratch slice): one per GROWING wave

// The workgroup's bookkeeping words (static LDS, file scope so that both instances of the problem code below share them)
__shared__ int g_next_comp, g_line_count, g_waves_done, g_pend_n;
#if LFG_EVAL_QUEUE
__shared__ grow::EvalQueue g_evalq;
#endif

// One problem.  BIG = false: its defined pixels all fit the LDS slice (every problem of the synthetic 640x480 frames): the
// context's "how many entries are in LDS" bounds are compile-time infinite, so the USED bits and x lists are plain LDS accesses
// -- with a run-time bound every access was `e < n ? lds[e] : hbm[e]`, which the compiler turns into ONE flat access through a
// selected address (the long way to LDS, and a wait for every global load in flight).  BIG = true: the same code with the
// bounds, for the problems that do not fit.  The kernel picks per workgroup (two launches, one per kind, run the two kinds one
// after the other: 11 % fewer frames/s on the real camera frames, whose problems are of both kinds).
//
// BM = true (round 4, k_lsd_grow_bm: the handle's busy-content form): no row starts and x lists at all -- the defined pixels as a
// bit plane of the scaled image plus a running count per 64-bit word (lsd_grow.h: a compact index is a rank), built here from the
// compact list; Hs * Ws / 64 * 9 bytes whatever the number of defined pixels (18.5 KB at 512 x 256), so a problem is "big"
// only when its USED bits do not fit (bm_used_cap entries), and those run the BIG code above in the same LDS.
// ZL = true (k_lsd_grow_zl, behind the bit-plane kernel): the bounded code with NOTHING in LDS -- row starts, x coordinates, USED
// bits and region lists all read from global memory.  Slow, and meant to be: it serves the problems beyond the bit-plane kernel's
// USED bits, which no frame measured has, and a launch of it asks for no dynamic LDS, so its (empty) workgroups pass through a busy
// chip in microseconds where the row-list slice made them queue for 1.4 ms per batch.
template <bool BIG, bool BM = false, bool ZL = false>
__device__ __forceinline__ void lsd_grow_problem(const LsdParams& p, const uint32_t* __restrict__ order,
                                                 const int* __restrict__ norder, const uint32_t* __restrict__ c_xy,
                                                 const float* __restrict__ c_deg, const double* __restrict__ c_mod,
                                                 const double* __restrict__ c_cs, const double* __restrict__ c_sn,
                                                 const int* __restrict__ row_start, const uint16_t* __restrict__ c_label,
                                                 const uint16_t* __restrict__ comp_list, const int* __restrict__ comp_count,
                                                 int comp_cap, uint32_t* reg, size_t reg_stride, uint32_t* gused,
                                                 float* tmp_lines, int* tmp_tags, float* lines, int* counts, int reg_lds, int def_lds, int pc,
                                                 double* pend_rec, int* pend_tag, int* pend_count, int pend_cap)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    int& next_comp = g_next_comp; int& line_count = g_line_count; int& waves_done = g_waves_done; int& pend_n = g_pend_n;
#if LFG_EVAL_QUEUE
    grow::EvalQueue& evalq = g_evalq;
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t Ps = (size_t)p.Hs * p.Ws;
    // LDS carve: [row starts] [USED bits] [region lists, one per wave] [x lists (u16)]
    // BM:        [bit plane (u64)] [word counts (u16)] [USED bits] [region lists]
    const int bm_words = bitplane_words(Ps);                 // + the word rank_before(H * W) reads; whole pairs (lsd_bitplane.h)
    if (ZL) { reg_lds = 0; def_lds = 0; }
    const int* rows = ZL ? row_start + (size_t)pc * (p.Hs + 1) : reinterpret_cast<const int*>(lds);
    uint32_t* usedc = BM ? lds + 2 * bm_words + (((bm_words >> 1) + 1) >> 1) : lds + ((p.Hs + 2) & ~1);
    uint32_t* lreg = usedc + ((def_lds + 31) >> 5) + 1;
    uint16_t* lxs = reinterpret_cast<uint16_t*>(lreg + GROW_LISTS * reg_lds);
    unsigned long long* bits64 = reinterpret_cast<unsigned long long*>(lds);
    uint16_t* pref = reinterpret_cast<uint16_t*>(lds + 2 * bm_words);
    const int n_def = norder[pc];
    const uint32_t* gxy = c_xy + (size_t)pc * p.rec_cap;
    const int* grs = row_start + (size_t)pc * (p.Hs + 1);
    if (BM) {
        __shared__ int wave_tot[GROW_WAVES];
        bitplane_build<64 * GROW_WAVES>(lds, gxy, n_def, p.Ws, Ps, wave_tot);
    } else if (!ZL) {
        for (int i = tid; i <= p.Hs; i += 64 * GROW_WAVES) lds[i] = (uint32_t)grs[i];
        for (int i = tid; i < n_def && i < def_lds; i += 64 * GROW_WAVES) lxs[i] = (uint16_t)(gxy[i] & 0xffffu);
    }
    if (!ZL)
        for (int i = tid; i <= ((def_lds + 31) >> 5); i += 64 * GROW_WAVES) usedc[i] = 0u;
    uint32_t* gu = gused + (size_t)pc * ((Ps + 31) / 32);
    if (n_def > def_lds)
        for (int i = tid; i < (n_def + 31) / 32; i += 64 * GROW_WAVES) gu[i] = 0u;
    if (tid == 0) { next_comp = 0; line_count = 0; waves_done = 0; pend_n = 0; }
#if LFG_EVAL_QUEUE
    if (tid == 0) { evalq.tail = 0; evalq.head = 0; evalq.growers = GROW_WAVES; }
    if (tid < LFG_QN) evalq.seq[tid] = 0;
#endif
    __threadfence_block();
    __syncthreads();
    const int n_comp = comp_count[pc];
    const uint16_t* clist = comp_list + (size_t)pc * comp_cap;
    float* tl = tmp_lines + (size_t)pc * p.cap_lines * 4;
    int* tt = tmp_tags + (size_t)pc * p.cap_lines;
    static_assert(LFG_QN <= 64, "ring slots are initialised by the first lanes");
    const bool use_queue = LFG_EVAL_QUEUE != 0;
    grow::Ctx c;
    c.W = p.Ws; c.H = p.Hs;
#if LFG_EVAL_QUEUE
    c.q = &evalq;
#else
    c.q = nullptr;
#endif
    c.pend_rec = pend_rec ? pend_rec + (size_t)pc * pend_cap * 12 : nullptr;
    c.pend_tag = pend_tag ? pend_tag + (size_t)pc * pend_cap : nullptr;
    c.pend_n = &pend_n; c.pend_cap = pend_cap;
    c.rows = rows; c.lxs = lxs; c.gxy = gxy; c.def_lds = BM ? -1 : BIG ? def_lds : 0x7fffffff;
    c.bits64 = bits64; c.pref = pref;
    c.deg = c_deg + (size_t)pc * p.rec_cap;
    c.mod = c_mod + (size_t)pc * p.rec_cap;
    c.cs = c_cs + (size_t)pc * p.rec_cap * 2;                       // interleaved (cos, sin) pairs: lsd_grow.h cs_sn()
    c.sd = p.c_sd ? p.c_sd + (size_t)pc * p.rec_cap * 2 : nullptr;
    c.sn = c_sn + (size_t)pc * p.rec_cap * 2;
    c.usedc = usedc; c.gused = gu; c.used_lds = BIG ? def_lds : 0x7fffffff;
    // every wave has its own region list: reg_lds entries in LDS, the rest in its slice of the problem's scratch.
    // Problems too large for k_lsd_label's LDS (> label_items defined pixels) come as ONE component: wave 0 takes it
    // with the whole scratch, the other waves have nothing to do.
    const bool single = n_def > p.label_items;
    c.lreg = lreg + (wave < GROW_LISTS ? wave : 0) * reg_lds; c.reg_lds = reg_lds;      // the evaluating wave never grows: no list of its own
    c.greg = reg + (size_t)pc * reg_stride + (single || wave >= GROW_LISTS ? (size_t)0 : (size_t)wave * p.label_items);
    c.log_nt = p.log_nt; c.log_eps = p.log_eps; c.density_th = p.density_th;
    c.prec = p.prec; c.p = p.p; c.scale = p.scaled ? p.scale : 1.0;
    c.min_reg_size = p.min_reg_size; c.refine = p.refine;
    c.label = c_label + (size_t)pc * p.rec_cap;
    c.tags = tt; c.line_count = &line_count;
#ifdef LFG_STAMPS
    for (int k = 0; k < 32; ++k) c.stamps[k] = 0;
    unsigned long long tb0 = __builtin_readcyclecounter();
#endif
    // components largest first (k_lsd_label sorted them), next one to whichever wave is free
    for (;;) {
        if (single && wave != 0) break;
        if (use_queue && wave == GROW_WAVES - 1) break;       // the workgroup's last wave only evaluates (see detect())
        int k = 0;
        if (lane == 0) k = atomicAdd(&next_comp, 1);
        k = __builtin_amdgcn_readfirstlane(k);
        if (k >= n_comp) break;
        c.root = (int)clist[k];
        (void)grow::detect(c, order + (size_t)pc * p.rec_cap, n_def, tl, p.cap_lines);
    }
    // no more components for this wave: help with the deferred evaluations until every growing wave is done and the ring is
    // empty -- or until there has been nothing to take for a while (an idle wave gives its slot back; whatever is pushed
    // after that is taken by the waves that finish later, the last grower always drains the ring)
#if LFG_EVAL_QUEUE
    {
        if (lane == 0) atomicSub(&evalq.growers, 1);
        // the last wave stays for as long as anyone grows; a wave that has finished its components only drains what is there
        const bool stay = wave == GROW_WAVES - 1;
        for (;;) {
            grow::Rect r;
            int tag = 0;
            if (grow::eval_pop(c, r, tag)) { (void)grow::evaluate_region<true>(c, r, tag, tl, p.cap_lines, 0); continue; }
            int g = 0;
            if (lane == 0) g = __hip_atomic_load(&evalq.growers, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            g = __builtin_amdgcn_readfirstlane(g);
            if (g == 0) {
                int left = 0;
                if (lane == 0) left = __hip_atomic_load(&evalq.tail, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - __hip_atomic_load(&evalq.head, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (__builtin_amdgcn_readfirstlane(left) <= 0) break;
                continue;
            }
            if (!stay) break;
            __builtin_amdgcn_s_sleep(64);
        }
    }
#endif
#ifdef LFG_STAMPS
    if (lane == 0) {
        // diagnostic: park the phase totals of every wave in the (otherwise unused) tail of the region scratch
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(reg + (size_t)pc * reg_stride + reg_stride - 64 * GROW_WAVES) + 32 * wave;   // 32 u64 per wave
        for (int k = 0; k < 32; ++k) dbg[k] = c.stamps[k];
        dbg[24] = __builtin_readcyclecounter() - tb0;
        dbg[25] = (unsigned long long)norder[pc];
        dbg[26] = (unsigned long long)n_comp;
    }
#endif
    // A wave that has run out of components leaves at once (a wave parked at a barrier would keep its registers and
    // with them a wave slot of its SIMD); the LAST wave to finish puts the lines back into the sequential order: a
    // line's place is the number of lines whose seed comes earlier in the seed list.
    __threadfence_block();
    int done = 0;
    if (lane == 0) done = atomicAdd(&waves_done, 1);
    done = __builtin_amdgcn_readfirstlane(done);
    if (done != GROW_WAVES - 1) return;
    __threadfence_block();
    const int total = *(volatile int*)&line_count;
    const int n = total < p.cap_lines ? total : p.cap_lines;
    float* out = lines + (size_t)pc * p.cap_lines * 4;
    for (int i = lane; i < n; i += 64) {
        const int ti = tt[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += tt[j] < ti ? 1 : 0;
        const float4 v = *reinterpret_cast<const float4*>(tl + 4 * i);
        *reinterpret_cast<float4*>(out + 4 * rank) = v;
    }
    if (lane == 0) {
        counts[pc] = total;     // may exceed cap_lines: the host reports LF_ERR_CAPACITY
        if (pend_count) pend_count[pc] = *(volatile int*)&pend_n;
    }
}

// MODE 2: every problem, the code chosen per workgroup.  MODE 0 / 1: only the problems that fit / do not fit the slice (the other
// kind returns at once) -- launched one after the other when (nearly) every problem fits: the kernel that then does all the
// work carries one copy of the problem code instead of two (+1 - 2 % frames/s on the synthetic lane frames); with problems of
// both kinds in numbers the two launches would run the kinds one after the other (-11 % on real camera frames), so the host
// picks MODE 2 there (launch_lsd_grow).
template <int MODE>
__global__ __launch_bounds__(64 * GROW_WAVES) __attribute__((amdgpu_waves_per_eu(LFG_WAVES))) void k_lsd_grow(LsdParams p, const uint32_t* __restrict__ order,
                                                 const int* __restrict__ norder, const uint32_t* __restrict__ c_xy,
                                                 const float* __restrict__ c_deg, const double* __restrict__ c_mod,
                                                 const double* __restrict__ c_cs, const double* __restrict__ c_sn,
                                                 const int* __restrict__ row_start, const uint16_t* __restrict__ c_label,
                                                 const uint16_t* __restrict__ comp_list, const int* __restrict__ comp_count,
                                                 int comp_cap, uint32_t* reg, size_t reg_stride, uint32_t* gused,
                                                 float* tmp_lines, int* tmp_tags, float* lines, int* counts, int reg_lds, int def_lds, const int* __restrict__ perm,
                                                 double* pend_rec, int* pend_tag, int* pend_count, int pend_cap, int big_above)
{
    const int pc = perm ? perm[blockIdx.x] : (int)blockIdx.x;        // launch order: longest problems first (k_lsd_rank)
    const bool big = norder[pc] > big_above;                         // = def_lds; behind k_lsd_grow_bm: what that kernel left
    if (MODE == 0 && big) return;
    if (MODE == 1 && !big) return;
    if (MODE != 0 && big)
        lsd_grow_problem<true>(p, order, norder, c_xy, c_deg, c_mod, c_cs, c_sn, row_start, c_label, comp_list, comp_count, comp_cap, reg, reg_stride,
                               gused, tmp_lines, tmp_tags, lines, counts, reg_lds, def_lds, pc, pend_rec, pend_tag, pend_count, pend_cap);
    if (MODE != 1 && !big)
        lsd_grow_problem<false>(p, order, norder, c_xy, c_deg, c_mod, c_cs, c_sn, row_start, c_label, comp_list, comp_count, comp_cap, reg, reg_stride,
                                gused, tmp_lines, tmp_tags, lines, counts, reg_lds, def_lds, pc, pend_rec, pend_tag, pend_count, pend_cap);
}

// The bit-plane form: every problem whose USED bits fit (used_cap entries); the few that do not are left to k_lsd_grow<1>, launched
// right behind with big_above = used_cap (one copy of the problem code per kernel: see MODE above).
__global__ __launch_bounds__(64 * GROW_WAVES) __attribute__((amdgpu_waves_per_eu(LFG_WAVES))) void k_lsd_grow_bm(LsdParams p, const uint32_t* __restrict__ order,
                                                 const int* __restrict__ norder, const uint32_t* __restrict__ c_xy,
                                                 const float* __restrict__ c_deg, const double* __restrict__ c_mod,
                                                 const double* __restrict__ c_cs, const double* __restrict__ c_sn,
                                                 const int* __restrict__ row_start, const uint16_t* __restrict__ c_label,
                                                 const uint16_t* __restrict__ comp_list, const int* __restrict__ comp_count,
                                                 int comp_cap, uint32_t* reg, size_t reg_stride, uint32_t* gused,
                                                 float* tmp_lines, int* tmp_tags, float* lines, int* counts, int reg_lds, int used_cap,
                                                 const int* __restrict__ perm, double* pend_rec, int* pend_tag, int* pend_count, int pend_cap)
{
    const int pc = perm ? perm[blockIdx.x] : (int)blockIdx.x;
    if (norder[pc] > used_cap) return;
    lsd_grow_problem<false, true>(p, order, norder, c_xy, c_deg, c_mod, c_cs, c_sn, row_start, c_label, comp_list, comp_count, comp_cap, reg, reg_stride,
                                  gused, tmp_lines, tmp_tags, lines, counts, reg_lds, used_cap, pc, pend_rec, pend_tag, pend_count, pend_cap);
}

__global__ __launch_bounds__(64 * GROW_WAVES) __attribute__((amdgpu_waves_per_eu(LFG_WAVES))) void k_lsd_grow_zl(LsdParams p, const uint32_t* __restrict__ order,
                                                 const int* __restrict__ norder, const uint32_t* __restrict__ c_xy,
                                                 const float* __restrict__ c_deg, const double* __restrict__ c_mod,
                                                 const double* __restrict__ c_cs, const double* __restrict__ c_sn,
                                                 const int* __restrict__ row_start, const uint16_t* __restrict__ c_label,
                                                 const uint16_t* __restrict__ comp_list, const int* __restrict__ comp_count,
                                                 int comp_cap, uint32_t* reg, size_t reg_stride, uint32_t* gused,
                                                 float* tmp_lines, int* tmp_tags, float* lines, int* counts, int big_above,
                                                 const int* __restrict__ perm, double* pend_rec, int* pend_tag, int* pend_count, int pend_cap)
{
    const int pc = perm ? perm[blockIdx.x] : (int)blockIdx.x;
    if (norder[pc] <= big_above) return;
    lsd_grow_problem<true, false, true>(p, order, norder, c_xy, c_deg, c_mod, c_cs, c_sn, row_start, c_label, comp_list, comp_count, comp_cap, reg, reg_stride,
                                        gused, tmp_lines, tmp_tags, lines, counts, 0, 0, pc, pend_rec, pend_tag, pend_count, pend_cap);
}

// K_lsd_eval: rect_improve + final coordinates (grow::evaluate_pending) of every region on a problem's pending list, then
// the accepted ones into seed order.  One workgroup per problem in the same longest-first order; wave w takes regions
// w, w + waves, ... (one region = up to 26 rectangle scans of a few dozen rows: a few tens of thousands of cycles), so a
// problem with r pending regions takes r / waves of them instead of the r the single evaluating wave of the ring form
// went through.  A region's answer goes back into its own list entry (the line over the first 16 bytes of the
// rectangle, tag = -1 when the NFA test rejects it) from ALL lanes: no lane-dependent branch in the loop (see
// evaluate_pending).  Needs of a problem here: its row starts and x lists (LDS, as in the growing kernel) and the
// angle plane.
__global__ __launch_bounds__(64 * LFG_EVAL_WAVES) void k_lsd_eval(LsdParams p, const int* __restrict__ norder, const uint32_t* __restrict__ c_xy,
                                                                   const float* __restrict__ c_deg, const int* __restrict__ row_start,
                                                                   double* pend_rec, int* pend_tag, const int* __restrict__ pend_count,
                                                                   int pend_cap, float* lines, int* counts, int def_lds,
                                                                   const int* __restrict__ perm, uint32_t* dbg_reg, size_t dbg_stride)
{
    extern __shared__ uint32_t lds[];
    __shared__ int waves_done;
    const int pc = perm ? perm[blockIdx.x] : (int)blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_pend = pend_count[pc];
    if (n_pend == 0) return;                                    // counts[pc] = 0 was written by the growing kernel
    if (n_pend > pend_cap) {                                    // more finished regions than the list holds: a capacity error of the run
        if (tid == 0) counts[pc] = p.cap_lines + 1;
        return;
    }
    int* rows = reinterpret_cast<int*>(lds);
    uint16_t* lxs = reinterpret_cast<uint16_t*>(lds + ((p.Hs + 2) & ~1));
    const int n_def = norder[pc];
    const uint32_t* gxy = c_xy + (size_t)pc * p.rec_cap;
    const int* grs = row_start + (size_t)pc * (p.Hs + 1);
    for (int i = tid; i <= p.Hs; i += 64 * LFG_EVAL_WAVES) rows[i] = grs[i];
    for (int i = tid; i < n_def && i < def_lds; i += 64 * LFG_EVAL_WAVES) lxs[i] = (uint16_t)(gxy[i] & 0xffffu);
    if (tid == 0) waves_done = 0;
    __syncthreads();
    grow::Ctx c;
    c.W = p.Ws; c.H = p.Hs;
    c.q = nullptr;
    c.rows = rows; c.lxs = lxs; c.gxy = gxy; c.def_lds = def_lds;
    c.deg = c_deg + (size_t)pc * p.rec_cap;
    c.mod = nullptr; c.cs = nullptr; c.sn = nullptr; c.sd = nullptr; c.usedc = nullptr; c.gused = nullptr; c.used_lds = 0;
    c.lreg = nullptr; c.greg = nullptr; c.reg_lds = 0;
    c.log_nt = p.log_nt; c.log_eps = p.log_eps; c.density_th = p.density_th;
    c.prec = p.prec; c.p = p.p; c.scale = p.scaled ? p.scale : 1.0;
    c.min_reg_size = p.min_reg_size; c.refine = p.refine;
    c.label = nullptr; c.root = 0;
    c.tags = nullptr; c.line_count = nullptr;
    c.pend_rec = nullptr; c.pend_tag = nullptr; c.pend_n = nullptr; c.pend_cap = 0;
#ifdef LFG_STAMPS
    for (int k = 0; k < 24; ++k) c.stamps[k] = 0;
    const unsigned long long tb0 = __builtin_readcyclecounter();
#endif
    double* pr = pend_rec + (size_t)pc * pend_cap * 12;
    int* pt = pend_tag + (size_t)pc * pend_cap;
    for (int k = wave; k < n_pend; k += LFG_EVAL_WAVES) {
        double* d = pr + (size_t)k * 12;
        grow::Rect r;
        r.x1 = d[0]; r.y1 = d[1]; r.x2 = d[2]; r.y2 = d[3]; r.width = d[4]; r.x = d[5]; r.y = d[6];
        r.theta = d[7]; r.dx = d[8]; r.dy = d[9]; r.prec = d[10]; r.p = d[11];
        const int tag = pt[k];
        float4 line = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool ok = grow::evaluate_pending(c, r, line);
        *reinterpret_cast<float4*>(d) = line;                   // the same 16 bytes from every lane
        pt[k] = ok ? tag : -1;
    }
#ifdef LFG_STAMPS
    if (lane == 0) {
        // diagnostic: the evaluating waves' totals in front of the growing waves' (see k_lsd_grow)
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(dbg_reg + (size_t)pc * dbg_stride + dbg_stride - 64 * GROW_WAVES - 64 * LFG_EVAL_WAVES) + 32 * wave;
        for (int k = 0; k < 24; ++k) dbg[k] = c.stamps[k];
        dbg[24] = __builtin_readcyclecounter() - tb0;
        dbg[25] = (unsigned long long)n_pend;
    }
#endif
    // the last wave to finish puts the accepted lines into the sequential order: a line's place is the number of accepted
    // lines whose seed comes earlier in the seed list
    __threadfence();
    int done = 0;
    if (lane == 0) done = atomicAdd(&waves_done, 1);
    done = __builtin_amdgcn_readfirstlane(done);
    if (done != LFG_EVAL_WAVES - 1) return;
    __threadfence();
    float* out = lines + (size_t)pc * p.cap_lines * 4;
    int total = 0;
    for (int i = lane; i < n_pend; i += 64) {
        const int ti = __builtin_nontemporal_load(pt + i);
        if (ti < 0) continue;
        int rank = 0;
        for (int j = 0; j < n_pend; ++j) { const int tj = __builtin_nontemporal_load(pt + j); rank += (tj >= 0 && tj < ti) ? 1 : 0; }
        ++total;
        if (rank < p.cap_lines) {
            const float* v = reinterpret_cast<const float*>(pr + (size_t)i * 12);
            *reinterpret_cast<float4*>(out + 4 * rank) = make_float4(__builtin_nontemporal_load(v), __builtin_nontemporal_load(v + 1),
                                                                      __builtin_nontemporal_load(v + 2), __builtin_nontemporal_load(v + 3));
        }
    }
    total = grow::wave_sum_i(total);
    if (lane == 0) counts[pc] = total;     // may exceed cap_lines: the host reports LF_ERR_CAPACITY
}

int lsd_grow_pend_cap(const LsdParams& p) { return LFG_EVAL_KERNEL ? 2 * p.cap_lines : 0; }

size_t lsd_grow_reg_stride(const LsdParams& p)
{
    // region scratch per problem: the whole scaled image for a single-component problem (one wave), or one
    // slice per wave -- components come from k_lsd_label only for problems of <= label_items defined pixels
    const size_t Ps = (size_t)p.Hs * p.Ws;
    // (a region holds defined pixels: no more than the problem's lists, rec_cap)
    const size_t need = (size_t)GROW_LISTS * (p.label_items_max > p.label_items ? p.label_items_max : p.label_items);
    const size_t all = (size_t)p.rec_cap < Ps ? (size_t)p.rec_cap : Ps;
    return all > need ? all : need;
}

// LDS per problem: row starts + (per defined pixel: 2 B of x + 1 USED bit) + one region-list head per wave.
// (An x-bucket table for the pixel searches -- 8 buckets per row, 9 KB -- was measured at +-0: the searches are
// ~6 % of the kernel and the table costs LDS residency.)
// lds_kb serves the 640x480 geometries (a few thousand defined pixels per problem).  Larger LSD images
// (1080p: 1536x576) have proportionally more defined pixels and far fewer problems per batch, so latency
// matters more than residency: grow the slice with the image (~3 % of the pixels defined), up to 64 KB.
static void lsd_grow_slice(const LsdParams& p, int lds_kb, int& reg_lds, int& def_lds, size_t& lds)
{
    const size_t Ps = (size_t)p.Hs * p.Ws;
    const size_t fixed = (size_t)((p.Hs + 2) & ~1) * 4;
    reg_lds = LFG_REG_LDS;
    const size_t regs = (size_t)GROW_LISTS * reg_lds * 4;
    size_t budget = (size_t)lds_kb * 1024 + regs - (size_t)reg_lds * 4;
    {
        const size_t want = fixed + regs + 8 + (size_t)((double)Ps * 0.03 * 17.0 / 8.0);
        if (want > budget) budget = want < (size_t)64 * 1024 ? want : (size_t)64 * 1024;
    }
    long long left = (long long)budget - (long long)fixed - (long long)regs - 8;
    def_lds = left > 0 ? (int)(left * 8 / 17) : 0;         // 2 B + 1/8 B per entry
    def_lds &= ~31;
    if ((size_t)def_lds > Ps) def_lds = (int)((Ps + 31) & ~(size_t)31);
    lds = fixed + (size_t)(((def_lds + 31) >> 5) + 1) * 4 + regs + (size_t)def_lds * 2 + 8;
}

// The bit-plane form's LDS: 9 bytes per 64 pixels (a 64-bit word of the plane, a 16-bit count per pair of words) + USED bits for used_cap entries + the region-list heads.  Only for images whose
// plane leaves room for problems to share a CU (<= 40 KB: 640x480 at both scales); false = use the row-list kernels.
static bool lsd_grow_bitmap_slice(const LsdParams& p, int reg_lds, bool busy, int used_override, int& used_cap, size_t& lds)
{
    const size_t Ps = (size_t)p.Hs * p.Ws;
    const size_t words = ((Ps >> 6) + 2) & ~(size_t)1;
    const size_t plane = words * 8 + (((words >> 1) + 1) >> 1) * 4;
    // USED bits for as many entries as would keep SIX problems on a CU (160 KB / 6, less 2.25 KB: 1.7 KB of static words and ring +
    // allocation granularity) -- the kernel runs five per CU since it went to 96 VGPRs (LFG_WAVES), and the sixth share is what
    // other kernels' workgroups find free on every CU (all 32 k entries the u16 counts allow, 0.6 KB more: measured equal to
    // slightly worse; 512 list entries per wave in LDS instead of 256: equal).  27.7 k entries at
    // 512 x 256: camera frames and the clutter frames have problems of 20 - 25 k defined pixels, and one of those in the row-list
    // code holds its batch up (16 k entries: -20 % frames/s there).  The counts are u16.
    (void)busy;
    const size_t lists = (size_t)GROW_LISTS * reg_lds * 4;
    const long long room = (long long)(160 * 1024 / 6 - 2304) - (long long)plane - (long long)lists - 12;
    used_cap = room > 512 ? (int)((room * 8) & ~31ll) : 4096;
    if (used_cap > 32768) used_cap = 32768;
    if (used_override > 0) used_cap = (used_override > 32768 ? 32768 : used_override) & ~31;   // tests: a small capacity sends problems down the overflow path (clamped: the plane's running counts are u16)
    if ((size_t)used_cap > Ps) used_cap = (int)((Ps + 31) & ~(size_t)31);
    lds = plane + (size_t)((used_cap >> 5) + 1) * 4 + lists + 8;
    return plane <= (size_t)28 * 1024 && Ps < ((size_t)1 << 21);
}

int lsd_grow_def_lds(const LsdParams& p, int lds_kb)
{
    int reg_lds, def_lds;
    size_t lds;
    lsd_grow_slice(p, lds_kb, reg_lds, def_lds, lds);
    return def_lds;
}

// lds_kb: the slice size.  13 KB (LFG_LDS_KB) holds every problem of the synthetic lane frames and gives the most resident
// problems; real camera frames have two to three times the edge pixels, many of their problems overflow 13 KB into the
// bounded (flat-access, HBM-backed) path, and a 28 KB slice is worth +13 - 30 % frames/s there (-7 % on the synthetic frames;
// 20 KB: half of that, 40 KB: no more) -- so the host moves a handle between kGrowLdsKb[] by the share of overflowing problems
// it saw in the last batch.
void launch_lsd_grow(const LsdParams& p, int n_frames, const uint32_t* order, const int* norder, const uint32_t* c_xy,
                     const float* c_deg, const double* c_mod, const double* c_cs, const double* c_sn,
                     const int* row_start, const uint16_t* c_label, const uint16_t* comp_list, const int* comp_count, int comp_cap,
                     uint32_t* reg, uint32_t* gused, float* tmp_lines, int* tmp_tags, float* lines, int* counts, const int* perm,
                     double* pend_rec, int* pend_tag, int* pend_count, int lds_kb, bool mixed, int bitmap, hipStream_t s)
{
    int reg_lds, def_lds;
    size_t lds;
    lsd_grow_slice(p, lds_kb > 0 ? lds_kb : LFG_LDS_KB, reg_lds, def_lds, lds);
    int big_above = def_lds;
#define LF_GROW_LAUNCH(MODE)                                                                                                                   \
    hipLaunchKernelGGL(k_lsd_grow<MODE>, dim3(n_frames * 3), dim3(64 * GROW_WAVES), lds, s, p, order, norder, c_xy, c_deg, c_mod, c_cs,        \
                       c_sn, row_start, c_label, comp_list, comp_count, comp_cap, reg, lsd_grow_reg_stride(p), gused, tmp_lines,               \
                       tmp_tags, lines, counts, reg_lds, def_lds, perm, pend_rec, pend_tag, pend_count, lsd_grow_pend_cap(p), big_above)
    size_t bm_lds = 0;
    int bm_used = 0;
    if (bitmap && !LFG_EVAL_KERNEL && lsd_grow_bitmap_slice(p, reg_lds, lds_kb > LFG_LDS_KB, bitmap > 1 ? bitmap : 0, bm_used, bm_lds)) {
        hipLaunchKernelGGL(k_lsd_grow_bm, dim3(n_frames * 3), dim3(64 * GROW_WAVES), bm_lds, s, p, order, norder, c_xy, c_deg, c_mod, c_cs,
                           c_sn, row_start, c_label, comp_list, comp_count, comp_cap, reg, lsd_grow_reg_stride(p), gused, tmp_lines,
                           tmp_tags, lines, counts, reg_lds, bm_used, perm, pend_rec, pend_tag, pend_count, lsd_grow_pend_cap(p));
        // what it left (none, on every frame measured): the bounded code with its tables in global memory (k_lsd_grow_zl)
        hipLaunchKernelGGL(k_lsd_grow_zl, dim3(n_frames * 3), dim3(64 * GROW_WAVES), 0, s, p, order, norder, c_xy, c_deg, c_mod, c_cs,
                           c_sn, row_start, c_label, comp_list, comp_count, comp_cap, reg, lsd_grow_reg_stride(p), gused, tmp_lines,
                           tmp_tags, lines, counts, bm_used, perm, pend_rec, pend_tag, pend_count, lsd_grow_pend_cap(p));
        (void)big_above;
        return;
    }
    if (mixed) { LF_GROW_LAUNCH(2); }
    else { LF_GROW_LAUNCH(0); LF_GROW_LAUNCH(1); }
#undef LF_GROW_LAUNCH
#if LFG_EVAL_KERNEL
    if (p.refine >= 2) {
        const size_t elds = (size_t)((p.Hs + 2) & ~1) * 4 + (size_t)def_lds * 2 + 8;
        hipLaunchKernelGGL(k_lsd_eval, dim3(n_frames * 3), dim3(64 * LFG_EVAL_WAVES), elds, s, p, norder, c_xy, c_deg, row_start, pend_rec,
                           pend_tag, pend_count, lsd_grow_pend_cap(p), lines, counts, def_lds, perm, reg, lsd_grow_reg_stride(p));
    }
#endif
}

}  // namespace lf
