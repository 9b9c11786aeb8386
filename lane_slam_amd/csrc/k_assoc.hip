// Associator (a-10 / a-11): nearest neighbour of every query descriptor in the live map.
//
// Reference semantics: BinaryDescriptorMatcher::match
// (/root/reference/src/line_descriptor/src/binary_descriptor_matcher.cpp:197-254) with
// Mihasher(256, 32), K = 1 (:635-753) and the popcount Hamming distance of
// bitops_custom.hpp:83-96: the exact Hamming nearest neighbour; candidates farther than
// D = 128 are never reported (:721).  The reference's multi-index hash is a CPU
// pointer-chasing structure rebuilt on every call; on MI355X the N x M distance matrix is one dense matrix-core
// contraction of +-1 operands: dot = 256 - 2 * hamming, exactly.
//
// Kernels (one launch per association; DESIGN.md section 4 and 5 have the measurements):
//   k_assoc_fp4 / k_assoc_fp4_gated   the product path: every code bit is an e2m1 nibble (+1.0 / -1.0), 64 bits per
//       v_mfma_scale_f32_32x32x64_f8f6f4, query side scaled by 2^9 through the E8M0 block scale, exact f32 accumulation;
//       the 32-column block number t enters through a fifth matrix step (weights 64, 8, 1 x minus the octal digits of
//       t), colour gating through one more (-147 456 when the colours differ): key = 512 * dot - t - penalty orders
//       candidates by distance, then by column block, so the running arg-max is ONE v_max_f32 per accumulator register.
//       Map rows: 128 bytes, blocked by 64-row tile (assoc_map_offset_fp4); colour rows: 32 bytes per map row.
//   k_assoc / k_assoc_plain           the int8 form (LF_ASSOC_INT8=1, kept for A/B runs): query bits +-32, map bits +-16,
//       v_mfma_i32_32x32x32_i8, K = 256; the block number is the chain's start value (plain) or, with gating, part of
//       a ninth step (K = 32: [16, 1] x [-(t >> 4), -(t & 15)], and 127 x -127 over ten bytes per colour group =
//       -161 290 when the colours differ).  Map rows: 256 bytes blocked by tile (assoc_map_offset) + 32-byte ninth-step rows.
//   All of them: 256 queries per workgroup (4 waves x 2 x 32 rows), query operands expanded from the raw 32-byte codes
//   into registers, map chunk streamed by LDS-DMA through tile buffers, hand-scheduled tile loop (k_assoc_loop.inc,
//   tools/gen_assoc_loop.py), per-chunk key rows written through and merged by the last workgroup to arrive; ties
//   resolve to the lowest map index.  Algorithmic ops: 2*N*M*256.
//   k_assoc_float: 72-d float LBD, Euclidean, on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain).
#include <cstdlib>
#include <cstdio>
#include "common.h"
#include "k_assoc_loop.inc"
// The tile loops set m0 before every global_load_lds and say so in their clobber lists, so that the compiler's
// merging of its own m0 set-ups never reaches across an asm statement; clang notes that m0 is a reserved register.
#pragma clang diagnostic ignored "-Winline-asm"

namespace lf {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

#ifdef LF_ASSOC_STAMPS
__device__ unsigned long long g_assoc_stamps[8 * 8192];
#define LF_STAMP(k) do { if (threadIdx.x == 0) { const unsigned w_ = blockIdx.y * gridDim.x + blockIdx.x; if (w_ < 8192) g_assoc_stamps[w_ * 8 + (k)] = (k) == 0 ? wall_clock64() : __builtin_readcyclecounter(); } } while (0)
#else
#define LF_STAMP(k) do { } while (0)
#endif
constexpr int AQW = 256;      // queries per workgroup (4 waves x 64)
constexpr int AM = 64;         // map entries per LDS tile
constexpr int kMaxBlocksPerChunk = 512;   // the in-accumulator block counter t has 9 bits
#ifndef LF_ASSOC_SMALL_MAX
#define LF_ASSOC_SMALL_MAX 12288
#endif
constexpr int kAssocSmallMax = LF_ASSOC_SMALL_MAX;   // associations of up to this many queries take the small shape (launch_assoc_core)

// Packs map rows for lf_associate's raw-map form (the live map keeps its rows packed: k_map.hip).  One thread per
// (row, code byte): 8 int8 = 2 dwords; thread 0 of a row also writes the row's ninth-step operand (zero counter bytes,
// -127 in the other colours' groups; colour >= 3 or no colours: matches every colour).
__global__ void k_assoc_pack_map(const uint8_t* __restrict__ codes, const uint8_t* __restrict__ colors, int n, int n_pad, int fp4,
                                 int8_t* __restrict__ out, int8_t* __restrict__ outc)
{
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t total = (size_t)n_pad * 32;
    if (t >= total) return;
    size_t row = t >> 5;
    uint32_t lo = 0, hi = 0;
    const bool live = row < (size_t)n;
    if (live) {
        uint32_t b = codes[t];
        // nibble -> 4 bytes of 0/1, then 0 -> +16, 1 -> -16
        uint32_t w0 = ((b & 15u) * 0x00204081u) & 0x01010101u;
        uint32_t w1 = ((b >> 4) * 0x00204081u) & 0x01010101u;
        lo = (w0 * 0xE0u) ^ 0x10101010u; hi = (w1 * 0xE0u) ^ 0x10101010u;
    }
    if (fp4) {
        // one e2m1 nibble per bit (padding rows: zeros = the value 0.0, "distance 128")
        *reinterpret_cast<uint32_t*>(out + assoc_map_offset_fp4(row, (int)(t & 31) * 4)) = live ? assoc_fp4_expand(codes[t]) : 0u;
        return;
    }
    *reinterpret_cast<uint2*>(out + assoc_map_offset(row, (int)(t & 31) * 8)) = make_uint2(lo, hi);
    if ((t & 31) == 0) {
        uint32_t w[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        if (live) {
            const int c = colors ? colors[row] : 255;
            uint8_t* wb = reinterpret_cast<uint8_t*>(w);
            if (c < 3)
                for (int g = 0; g < 3; ++g)
                    if (g != c) for (int k = 0; k < 10; ++k) wb[2 + 10 * g + k] = (uint8_t)(-127);
        }
        uint4* o = reinterpret_cast<uint4*>(outc + row * 32);
        o[0] = make_uint4(w[0], w[1], w[2], w[3]);
        o[1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
}

// four code bits -> four int8 query operand bytes: 0 -> +32, 1 -> -32
__device__ __forceinline__ int q_expand(uint32_t nibble)
{
    return (int)((((nibble & 15u) * 0x00204081u) & 0x01010101u) * 0xC0u ^ 0x20202020u);
}

template <int QW>
__device__ __forceinline__ void assoc_publish_and_merge(unsigned int mine, int mine_q, int nq, int max_distance, unsigned int* __restrict__ part,
                                                        int* __restrict__ done, int32_t* __restrict__ idx, float* __restrict__ dist,
                                                        int* __restrict__ tie_pieces, int* __restrict__ tie_counts, unsigned long long* __restrict__ tie_res)
{
    // The chunk's 256 keys go to this workgroup's row of part[] -- NOT atomicMin on a shared word per query: device-scope
    // atomics are executed at the memory side, 131 072 of them (16 k queries x 8 chunks) are slow however short the
    // chunks were.  The row is stored write-through (agent-scope relaxed atomic stores = sc1), drained, and then ONE
    // atomic per workgroup counts the arrival: no release fence (an agent-scope __threadfence() writes the L2 back and was
    // measured at 20-30 us per workgroup here).  The last workgroup of a query block to arrive reads the rows of all
    // chunks with sc1 loads, writes idx / dist for its 256 queries and puts the counter back to zero, so an association is
    // ONE kernel launch: no init pass, no finish pass.
    // (QW = queries per workgroup: 256, or 128 in the small shape -- there only the first 128 threads carry a query, and mine_q < QW)
    if (QW == AQW || (int)threadIdx.x < QW)
        __hip_atomic_store(part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * QW + mine_q, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __shared__ int s_last;
    __syncthreads();
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(done + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.y - 1;
    __syncthreads();
    LF_STAMP(5);
    if (!s_last) return;
    const bool has_q = QW == AQW || (int)threadIdx.x < QW;
    const int qq = has_q ? blockIdx.x * QW + (int)threadIdx.x : nq;       // (threads without a query: beyond the end)
    int best_ham = -1;                                                     // this query's reported distance (-1: none)
    if (qq < nq) {
        // eight rows in flight per thread: the merge is the serial tail of the launch (the last workgroup of the query block
        // runs it alone), a dependent load per chunk would cost a memory round trip each
        unsigned int v = 0x7fffffffu;
        const unsigned int* col = part + (size_t)blockIdx.x * QW + threadIdx.x;
        const size_t stride = (size_t)gridDim.x * QW;
        const unsigned int nc = gridDim.y;
        unsigned int c = 0;
        for (; c + 8 <= nc; c += 8) {
            unsigned int t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = __hip_atomic_load(col + (size_t)(c + k) * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int k = 0; k < 8; ++k) v = min(v, t[k]);
        }
        for (; c < nc; ++c) v = min(v, __hip_atomic_load(col + (size_t)c * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const int ham = (int)(v >> 22);
        if (v == 0x7fffffffu || ham > max_distance) { idx[qq] = -1; dist[qq] = -1.f; }
        else { idx[qq] = (int)(v & 0x1fffffu); dist[qq] = (float)ham; best_ham = ham; }
    }
    if (tie_pieces) {
        // For the tie pass (k_assoc_ties.hip), in the same breath: per map chunk, the queries of this block whose best distance in the
        // chunk IS their best distance overall -- a candidate as near as the optimum can sit nowhere else.  Wave w of the block
        // writes them as piece (chunk, 4 blockIdx.x + w): up to 64 query numbers and a count (no atomics, no kernel of its own).
        // (the tie pass counts pieces per 256-query block, four each: the small shape's 128-query workgroups write two, and its last
        // workgroup also the empty ones an odd number of them leaves)
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const int n_pieces = QW == AQW ? (int)gridDim.x * 4 : ((nq + AQW - 1) / AQW) * 4;
        const int piece = blockIdx.x * (QW / 64) + wv;
        const bool writes_piece = wv < QW / 64 || (blockIdx.x == gridDim.x - 1 && piece < n_pieces);
        if (qq < nq) tie_res[qq] = ~0ull;
        const unsigned int* col = part + (size_t)blockIdx.x * QW + threadIdx.x;
        const size_t stride = (size_t)gridDim.x * QW;
        const unsigned int nc = gridDim.y;
        for (unsigned int c0 = 0; c0 < nc; c0 += 8) {
            unsigned int t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = (c0 + k < nc && qq < nq) ? __hip_atomic_load(col + (size_t)(c0 + k) * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x7fffffffu;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned int c = c0 + k;
                if (c >= nc) break;
                // (a query whose optimum is distance 0 needs no second look: every candidate at distance 0 equals the query, all their
                // discovery keys are (weight 0, substring 0, place 0), and the train index decides -- the lowest, which this pass reports)
                const bool on = best_ham > 0 && t[k] != 0x7fffffffu && (int)(t[k] >> 22) == best_ham;
                const unsigned long long bo = __ballot(on);
                if (lane == 0 && writes_piece) tie_counts[(size_t)c * n_pieces + piece] = __popcll(bo);
                if (on) tie_pieces[((size_t)c * n_pieces + piece) * 64 + __popcll(bo & ((1ull << lane) - 1ull))] = qq;
            }
        }
    }
    if (threadIdx.x == 0) __hip_atomic_store(done + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One workgroup: 256 queries (4 waves x 2 x 32 rows) against one chunk of the map.
//  * The query operands are expanded from the raw 32-byte codes straight into the A-fragment registers (no packed query
//    array in memory, no pack kernel): lane (r32, half) of a wave holds, for row block b and step s, the 16 bits
//    [32 s + 16 half, +16) of query q0 + 32 b + r32 as 16 int8.
//  * The map chunk streams through three 16 KB LDS buffers; the tile loop is hand-scheduled assembly
//    (k_assoc_loop.inc, generated by tools/gen_assoc_loop.py: register map, schedule, hazard notes there).  It leaves
//    the running keys in LDS.
//  * Keys are decoded and reduced to one word per query and chunk; the merge over the chunks is at the end of the body.
template <bool GATED>
__device__ __forceinline__ void assoc_body(const uint8_t* __restrict__ q, const uint8_t* __restrict__ qcolor, int nq,
                                           const int8_t* __restrict__ mx, const int8_t* __restrict__ mcx,
                                           int nm_bound, const int* __restrict__ nm_dev, int nm_pad, int m_chunk,
                                           int max_distance, unsigned int* __restrict__ part, int* __restrict__ done,
                                           int32_t* __restrict__ idx, float* __restrict__ dist, int8_t* tile, int8_t* ctile, uint2* xtab,
                                           int* __restrict__ tie_pieces, int* __restrict__ tie_counts, unsigned long long* __restrict__ tie_res)
{
    // nm_bound sized the grid on the host; when the exact size is only known on the device (the live map's size
    // after an update still in flight) it is read here.  Rows in [size, bound) are all-zero operands and are dropped
    // below exactly like padding rows, so the result does not depend on how loose the bound was.
    LF_STAMP(0); LF_STAMP(1);
    const int nm = nm_dev ? min(nm_bound, *nm_dev) : nm_bound;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q0 = blockIdx.x * AQW + wave * 64;
    const int r32 = lane & 31, half = lane >> 5;
    const int m_begin = blockIdx.y * m_chunk;
    const int m_end = min(nm_pad, m_begin + m_chunk);
    const int n_tiles = __builtin_amdgcn_readfirstlane((m_end - m_begin) / AM);
    unsigned int mine = 0x7fffffffu;            // this thread's query of the block (mine_q) against this chunk
    int mine_q = threadIdx.x;
    if (n_tiles > 0) {
        // byte -> eight int8 operand bytes (bit j of the byte -> +32 / -32), one table for the workgroup: the expansion
        // of the 2 x 8 fragments is then two 8-byte LDS reads each instead of ~20 vector instructions each (code size
        // matters here: this straight-line code runs once, and every cold instruction line is a fetch stall)
        if (threadIdx.x < 256) {
            const uint32_t t = threadIdx.x;
            xtab[t] = make_uint2((uint32_t)q_expand(t), (uint32_t)q_expand(t >> 4));
        }
        __syncthreads();
        v4i A[2][8], AX[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int qi = q0 + 32 * b + r32;
            uint4 c0 = make_uint4(0, 0, 0, 0), c1 = c0;        // padding rows: any operand will do, they are never reported
            if (qi < nq) {
                c0 = *reinterpret_cast<const uint4*>(q + (size_t)qi * 32);
                c1 = *reinterpret_cast<const uint4*>(q + (size_t)qi * 32 + 16);
            }
            const uint32_t d[8] = { c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w };
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const uint32_t hw = d[s] >> (16 * half);               // code bytes 4 s + 2 half, + 1
                const uint2 e0 = xtab[hw & 0xffu], e1 = xtab[(hw >> 8) & 0xffu];
                A[b][s] = v4i{ (int)e0.x, (int)e0.y, (int)e1.x, (int)e1.y };
            }
            if (GATED) {
                // ninth-step operand of a query: [16, 1] counter weights, then 127 in the ten bytes of ITS colour's group
                const int c = qi < nq ? (int)qcolor[qi] : 255;
                const int g0 = c == 0 ? 0x7f7f7f7f : 0, g1 = c == 1 ? 0x7f7f7f7f : 0, g2 = c == 2 ? 0x7f7f7f7f : 0;
                if (half == 0) AX[b] = v4i{ (qi < nq ? 0x0110 : 0) | (g0 & (int)0x7f7f0000), g0, g0, g1 };
                else AX[b] = v4i{ g1, (g1 & 0x00007f7f) | (g2 & (int)0x7f7f0000), g2, g2 };
            }
        }
        const uint32_t lds_tile = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)tile;
        const uint32_t vfrag = lds_tile + half * 1024 + r32 * 16;
        const uint32_t voff = (uint32_t)lane * 16u;
        const uint32_t vdump = lds_tile + wave * 8192 + lane * 16;
        const uint64_t mbase = (uint64_t)(size_t)(mx + (size_t)m_begin * 256 + wave * 1024);
        const uint32_t mlo = __builtin_amdgcn_readfirstlane((uint32_t)mbase), mhi = __builtin_amdgcn_readfirstlane((uint32_t)(mbase >> 32));
        const uint32_t m0base = __builtin_amdgcn_readfirstlane(lds_tile + wave * 1024);
        LF_STAMP(2);
#define LF_A_OPERANDS \
        [a00] "{v[40:43]}"(A[0][0]), [a01] "{v[44:47]}"(A[0][1]), [a02] "{v[48:51]}"(A[0][2]), [a03] "{v[52:55]}"(A[0][3]), \
        [a04] "{v[56:59]}"(A[0][4]), [a05] "{v[60:63]}"(A[0][5]), [a06] "{v[64:67]}"(A[0][6]), [a07] "{v[68:71]}"(A[0][7]), \
        [a10] "{v[72:75]}"(A[1][0]), [a11] "{v[76:79]}"(A[1][1]), [a12] "{v[80:83]}"(A[1][2]), [a13] "{v[84:87]}"(A[1][3]), \
        [a14] "{v[88:91]}"(A[1][4]), [a15] "{v[92:95]}"(A[1][5]), [a16] "{v[96:99]}"(A[1][6]), [a17] "{v[100:103]}"(A[1][7])
        if (GATED) {
            const uint32_t lds_ctile = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)ctile;
            const uint32_t vcfrag = lds_ctile + r32 * 32 + 16 * half;
            const uint64_t cbase = (uint64_t)(size_t)(mcx + (size_t)m_begin * 32 + (wave & 1) * 1024);
            const uint32_t clo = __builtin_amdgcn_readfirstlane((uint32_t)cbase), chi = __builtin_amdgcn_readfirstlane((uint32_t)(cbase >> 32));
            const uint32_t m0c = __builtin_amdgcn_readfirstlane(lds_ctile + (wave & 1) * 1024);
            const uint32_t tmaskv = half == 0 ? 0xffffu : 0u;          // the counter bytes live in k = 0, 1 (lanes 0..31)
            const uint64_t mpair = ((uint64_t)mhi << 32) | mlo, cpair = ((uint64_t)chi << 32) | clo;
            asm volatile(LF_ASSOC_LOOP_GATED
                         :
                         : LF_A_OPERANDS, [ax0] "{v[232:235]}"(AX[0]), [ax1] "{v[236:239]}"(AX[1]), [vfrag] "v"(vfrag),
                           [vcfrag] "v"(vcfrag), [voff] "v"(voff), [vdump] "v"(vdump), [tmaskv] "v"(tmaskv), [mbase] "s"(mpair),
                           [cbase] "s"(cpair), [m0base] "s"(m0base), [m0c] "s"(m0c), [ntiles] "s"(n_tiles)
                         : LF_ASSOC_LOOP_CLOBBERS_GATED);
        } else {
            const uint64_t mpair = ((uint64_t)mhi << 32) | mlo;
            asm volatile(LF_ASSOC_LOOP_PLAIN
                         :
                         : LF_A_OPERANDS, [vfrag] "v"(vfrag), [voff] "v"(voff), [vdump] "v"(vdump), [mbase] "s"(mpair),
                           [m0base] "s"(m0base), [ntiles] "s"(n_tiles)
                         : LF_ASSOC_LOOP_CLOBBERS_PLAIN);
        }
#undef LF_A_OPERANDS
        LF_STAMP(3);
        // key -> (distance, column): 512 * dot = ceil(key / 512) * 512, t = 512 * dot - key; this lane's column inside
        // block t is r32.  Padding columns (>= nm; their rows are zero, i.e. "distance 128") are dropped here: a padding
        // column can only have displaced candidates with a negative dot, which are beyond 128 and never reported.  A
        // candidate of another colour (gating) decodes to a distance beyond 128 and is dropped below.  Running key
        // i = 16 b + r of this lane sits in the wave's 8 KB of LDS.
        // 1) every lane decodes its 32 keys in place (column = 32 t + this lane's r32)
        int* dump = reinterpret_cast<int*>(tile) + wave * 2048;
#pragma unroll 2
        for (int g = 0; g < 8; ++g) {                                  // four keys per 16-byte access: no bank conflicts
            int4* slot = reinterpret_cast<int4*>(dump + g * 256 + lane * 4);
            const int4 k4 = *slot;
            const int key[4] = { k4.x, k4.y, k4.z, k4.w };
            int v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = 0x7fffffff;
                if (key[j] != (int)0x80000000) {
                    const int dot512 = (key[j] + 511) & ~511;
                    const int col = m_begin + 32 * (dot512 - key[j]) + r32;
                    if (col < nm) v[j] = (((256 << 9) - dot512) << 12) | col;      // hamming << 22 | col
                }
            }
            *slot = make_int4(v[0], v[1], v[2], v[3]);
        }
        __syncthreads();
        // 2) one lane per query: key i = 16 b + r of the 32 lanes (r32) of half h is query row (r & 3) + 8 (r >> 2) + 4 h
        //    of block b; lane (i, h) takes the minimum over r32, reading in an order rotated by i >> 2 (bank spread)
        {
            const int i = lane & 31, h = lane >> 5;
            const int* src = dump + (i >> 2) * 256 + h * 128 + (i & 3);
            int v = 0x7fffffff;
#pragma unroll 8
            for (int k = 0; k < 32; ++k) v = min(v, src[((k + (i >> 2)) & 31) * 4]);
            const int b = i >> 4, r = i & 15;
            mine = (unsigned int)v;
            mine_q = wave * 64 + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * h;
        }
    }
    LF_STAMP(4);
    assoc_publish_and_merge<AQW>(mine, mine_q, nq, max_distance, part, done, idx, dist, tie_pieces, tie_counts, tie_res);
}

// The ungated kernel on the FP4 matrix instruction (k_assoc_loop.inc, LF_ASSOC_LOOP_FP4; gen_assoc_loop.py gen_fp4 has the
// arithmetic): same work split and the same reduce / publish / merge tail as assoc_body; the map rows are e2m1 nibbles
// (128 bytes per row, 8 KB tiles), the queries are expanded from the raw codes through a byte -> 8 nibbles table.
template <bool GATED, int NRB = 2>
__device__ __forceinline__ void assoc_body_fp4(const uint8_t* __restrict__ q, const uint8_t* __restrict__ qcolor, int nq,
                                               const int8_t* __restrict__ mx, const int8_t* __restrict__ mcx,
                                               int nm_bound, const int* __restrict__ nm_dev, int nm_pad, int m_chunk,
                                               int max_distance, unsigned int* __restrict__ part, int* __restrict__ done,
                                               int32_t* __restrict__ idx, float* __restrict__ dist, int8_t* tile, int8_t* ctile,
                                               uint32_t* xtab, uint32_t* ttab,
                                               int* __restrict__ tie_pieces, int* __restrict__ tie_counts, unsigned long long* __restrict__ tie_res)
{
    LF_STAMP(0); LF_STAMP(1);
    const int nm = nm_dev ? min(nm_bound, *nm_dev) : nm_bound;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int QW = 128 * NRB;                 // NRB 32-query row blocks per wave (1: the small shape, k_assoc_fp4_s)
    const int q0 = blockIdx.x * QW + wave * 32 * NRB;
    const int r32 = lane & 31, half = lane >> 5;
    const int m_begin = blockIdx.y * m_chunk;
    const int m_end = min(nm_pad, m_begin + m_chunk);
    const int n_tiles = __builtin_amdgcn_readfirstlane((m_end - m_begin) / AM);
    unsigned int mine = 0x7fffffffu;
    int mine_q = NRB == 2 ? (int)threadIdx.x : wave * 32 + (lane & 31);
    if (n_tiles > 0) {
        xtab[threadIdx.x] = assoc_fp4_expand(threadIdx.x);
        // step five's map-side operand per block number t = 64 a + 8 b + c: minus the octal digits, each digit two e2m1 values
        // (0 1 2 3 4 4+1 6 4+3); k-half 0 carries a (weights 64) and b (weights 8), k-half 1 carries c (weights 1)
        for (int t = threadIdx.x; t < 512; t += 256) {
            const unsigned long long dig = 0xde0fae0e0d0c0a00ull;
            const uint32_t a = (uint32_t)(dig >> (8 * (t >> 6))) & 0xffu, b = (uint32_t)(dig >> (8 * ((t >> 3) & 7))) & 0xffu, c = (uint32_t)(dig >> (8 * (t & 7))) & 0xffu;
            ttab[t] = a | (b << 8);
            ttab[512 + t] = c;
        }
        __syncthreads();
        LF_STAMP(6);
        v4i A[2][4], AXC[2];
#pragma unroll
        for (int b = 0; b < NRB; ++b) {
            const int qi = q0 + 32 * b + r32;
            uint4 c0 = make_uint4(0, 0, 0, 0), c1 = c0;        // padding rows: any operand will do, they are never reported
            if (qi < nq) {
                c0 = *reinterpret_cast<const uint4*>(q + (size_t)qi * 32);
                c1 = *reinterpret_cast<const uint4*>(q + (size_t)qi * 32 + 16);
            }
            if (GATED) {
                // colour step, query side: 6.0 (e2m1 0x7) in the nibble of the query's colour, k-half 0 only
                const int c = qi < nq ? (int)qcolor[qi] : 255;
                AXC[b] = v4i{ (half == 0 && c < 3) ? (0x7 << (4 * c)) : 0, 0, 0, 0 };
            }
            // step s, k-half `half`: bits [64 s + 32 half, +32) = code dword 2 s + half.  (Selects on registers, not an indexed
            // array: the compiler turns `half ? d[2 s + 1] : d[2 s]` into d[2 s + half], keeps d[] in memory, promotes it to
            // LDS and then needs the workgroup size from the dispatch packet -- a scalar load from host memory, 15 us.)
            const uint32_t w4[4] = { half ? c0.y : c0.x, half ? c0.w : c0.z, half ? c1.y : c1.x, half ? c1.w : c1.z };
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const uint32_t w = w4[s];
                A[b][s] = v4i{ (int)xtab[w & 0xffu], (int)xtab[(w >> 8) & 0xffu], (int)xtab[(w >> 16) & 0xffu], (int)xtab[w >> 24] };
            }
        }
        LF_STAMP(7);
        // step five, query side: weights 4, 4, 0.5, 0.5 (block scale 2^4 -> 64, 64, 8, 8) in k-half 0; 1, 1 (scale 2^0) in k-half 1
        const v4i AX = v4i{ half ? 0x22 : 0x1166, 0, 0, 0 };
        const uint32_t scl5 = half ? 0x7f7f7f7fu : 0x83838383u;
        const uint32_t vtab = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)ttab + half * 2048;
        const uint32_t lds_tile = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)tile;
        const uint32_t vfrag = lds_tile + half * 1024 + r32 * 16;
        const uint32_t voff = (uint32_t)lane * 16u;
        const uint32_t vdump = lds_tile + wave * (4096 * NRB) + lane * 16;
        const uint64_t mbase = (uint64_t)(size_t)(mx + (size_t)m_begin * 128 + wave * 1024);
        const uint32_t mlo = __builtin_amdgcn_readfirstlane((uint32_t)mbase), mhi = __builtin_amdgcn_readfirstlane((uint32_t)(mbase >> 32));
        const uint64_t mpair = ((uint64_t)mhi << 32) | mlo;
        const uint32_t m0base = __builtin_amdgcn_readfirstlane(lds_tile + wave * 1024);
        LF_STAMP(2);
        if (GATED) {
            const uint32_t lds_ctile = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)ctile;
            const uint32_t vcfrag = lds_ctile + r32 * 32 + 16 * half;
            const uint64_t cbase = (uint64_t)(size_t)(mcx + (size_t)m_begin * 32 + (wave & 1) * 1024);
            const uint32_t clo = __builtin_amdgcn_readfirstlane((uint32_t)cbase), chi = __builtin_amdgcn_readfirstlane((uint32_t)(cbase >> 32));
            const uint64_t cpair = ((uint64_t)chi << 32) | clo;
            const uint32_t m0c = __builtin_amdgcn_readfirstlane(lds_ctile + (wave & 1) * 1024);
            if (NRB == 1)
            asm volatile(LF_ASSOC_LOOP_FP4_GATED_S
                         :
                         : [a00] "v"(A[0][0]), [a01] "v"(A[0][1]), [a02] "v"(A[0][2]), [a03] "v"(A[0][3]), [ax] "v"(AX), [scl5] "v"(scl5), [vtab] "v"(vtab),
                           [axc0] "v"(AXC[0]), [vcfrag] "v"(vcfrag), [cbase] "s"(cpair), [m0c] "s"(m0c),
                           [vfrag] "v"(vfrag), [voff] "v"(voff), [vdump] "v"(vdump), [mbase] "s"(mpair), [m0base] "s"(m0base), [ntiles] "s"(n_tiles)
                         : LF_ASSOC_LOOP_CLOBBERS_FP4_S);
            else
            asm volatile(LF_ASSOC_LOOP_FP4_GATED
                         :
                         : [a00] "v"(A[0][0]), [a01] "v"(A[0][1]), [a02] "v"(A[0][2]), [a03] "v"(A[0][3]), [a10] "v"(A[1][0]), [a11] "v"(A[1][1]),
                           [a12] "v"(A[1][2]), [a13] "v"(A[1][3]), [ax] "v"(AX), [scl5] "v"(scl5), [vtab] "v"(vtab), [axc0] "v"(AXC[0]),
                           [axc1] "v"(AXC[1]), [vcfrag] "v"(vcfrag), [cbase] "s"(cpair), [m0c] "s"(m0c),
                           [vfrag] "v"(vfrag), [voff] "v"(voff), [vdump] "v"(vdump), [mbase] "s"(mpair), [m0base] "s"(m0base), [ntiles] "s"(n_tiles)
                         : LF_ASSOC_LOOP_CLOBBERS_FP4);
        } else if (NRB == 1)
        asm volatile(LF_ASSOC_LOOP_FP4_S
                     :
                     : [a00] "v"(A[0][0]), [a01] "v"(A[0][1]), [a02] "v"(A[0][2]), [a03] "v"(A[0][3]), [ax] "v"(AX), [scl5] "v"(scl5), [vtab] "v"(vtab),
                       [vfrag] "v"(vfrag), [voff] "v"(voff), [vdump] "v"(vdump), [mbase] "s"(mpair), [m0base] "s"(m0base), [ntiles] "s"(n_tiles)
                     : LF_ASSOC_LOOP_CLOBBERS_FP4_S);
        else
        asm volatile(LF_ASSOC_LOOP_FP4
                     :
                     : [a00] "v"(A[0][0]), [a01] "v"(A[0][1]), [a02] "v"(A[0][2]), [a03] "v"(A[0][3]), [a10] "v"(A[1][0]), [a11] "v"(A[1][1]),
                       [a12] "v"(A[1][2]), [a13] "v"(A[1][3]), [ax] "v"(AX), [scl5] "v"(scl5), [vtab] "v"(vtab),
                       [vfrag] "v"(vfrag), [voff] "v"(voff), [vdump] "v"(vdump), [mbase] "s"(mpair), [m0base] "s"(m0base), [ntiles] "s"(n_tiles)
                     : LF_ASSOC_LOOP_CLOBBERS_FP4);
        LF_STAMP(3);
        // keys are floats here: 512 * dot - t, exact integers
        float* dump = reinterpret_cast<float*>(tile) + wave * (1024 * NRB);
#pragma unroll 2
        for (int g = 0; g < 4 * NRB; ++g) {                            // four keys per 16-byte access: no bank conflicts
            float4* slot = reinterpret_cast<float4*>(dump + g * 256 + lane * 4);
            const float4 k4 = *slot;
            const float kf[4] = { k4.x, k4.y, k4.z, k4.w };
            int v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = 0x7fffffff;
                if (kf[j] > -1.0e30f) {
                    const int key = (int)kf[j];
                    const int dot512 = (key + 511) & ~511;
                    const int col = m_begin + 32 * (dot512 - key) + r32;
                    if (col < nm) v[j] = (((256 << 9) - dot512) << 12) | col;      // hamming << 22 | col
                }
            }
            *reinterpret_cast<int4*>(slot) = make_int4(v[0], v[1], v[2], v[3]);
        }
        __syncthreads();
        if (NRB == 2) {
            const int i = lane & 31, h = lane >> 5;
            const int* src = reinterpret_cast<const int*>(dump) + (i >> 2) * 256 + h * 128 + (i & 3);
            int v = 0x7fffffff;
#pragma unroll 8
            for (int k = 0; k < 32; ++k) v = min(v, src[((k + (i >> 2)) & 31) * 4]);
            const int b = i >> 4, r = i & 15;
            mine = (unsigned int)v;
            mine_q = wave * 64 + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * h;
        } else {
            // one row block: 16 running registers x 2 k-halves = 32 queries per wave; the two halves of the wave split the 32 columns of
            // a query between them and meet through one DPP-free shuffle
            const int i = lane & 15, h = (lane >> 4) & 1, part2 = lane >> 5;
            const int* src = reinterpret_cast<const int*>(dump) + (i >> 2) * 256 + h * 128 + (i & 3);
            int v = 0x7fffffff;
#pragma unroll 8
            for (int k = 0; k < 16; ++k) v = min(v, src[((16 * part2 + ((k + (i >> 2)) & 15)) & 31) * 4]);
            v = min(v, __shfl_xor(v, 32));
            const int r = i;
            mine = (unsigned int)v;
            mine_q = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        }
    }
    // (small shape: the keys of the workgroup's 128 queries sit in the first 32 lanes of its four waves: hand them to threads 0 .. 127)
    if (NRB == 1) {
        __shared__ unsigned int s_keys[128];
        __syncthreads();
        if ((threadIdx.x & 63) < 32) s_keys[mine_q] = mine;
        __syncthreads();
        if (threadIdx.x < 128) { mine = s_keys[threadIdx.x]; mine_q = threadIdx.x; }
    }
    LF_STAMP(4);
    assoc_publish_and_merge<QW>(mine, mine_q, nq, max_distance, part, done, idx, dist, tie_pieces, tie_counts, tie_res);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_assoc_fp4(const uint8_t* __restrict__ q, int nq, const int8_t* __restrict__ mx,
                                               int nm_bound, const int* __restrict__ nm_dev, int nm_pad, int m_chunk, int max_distance,
                                               unsigned int* __restrict__ part, int* __restrict__ done, int32_t* __restrict__ idx, float* __restrict__ dist,
                                               int* __restrict__ tie_pieces, int* __restrict__ tie_counts, unsigned long long* __restrict__ tie_res)
{
    __shared__ __attribute__((aligned(1024))) int8_t tile[6 * AM * 128];     // six 8 KB tile buffers (gen_fp4 nbuf); the four key dumps reuse 32 KB of them
    __shared__ uint32_t xtab[256];
    __shared__ uint32_t ttab[1024];
    assoc_body_fp4<false>(q, nullptr, nq, mx, nullptr, nm_bound, nm_dev, nm_pad, m_chunk, max_distance, part, done, idx, dist, tile, nullptr, xtab, ttab, tie_pieces, tie_counts, tie_res);
}

// The SMALL shape (round 6): one row block per wave, three tile buffers -- 24 KB + tables and ~160 registers per workgroup, three waves per
// SIMD.  In a pipeline whose CUs are full of region-growing waves the big shape's workgroups (54 KB, 239 registers: three growing waves
// must leave every SIMD of a CU) wait four to ten times their own duration for room; this one runs longer alone and starts sooner.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_assoc_fp4_s(const uint8_t* __restrict__ q, int nq, const int8_t* __restrict__ mx,
                                               int nm_bound, const int* __restrict__ nm_dev, int nm_pad, int m_chunk, int max_distance,
                                               unsigned int* __restrict__ part, int* __restrict__ done, int32_t* __restrict__ idx, float* __restrict__ dist,
                                               int* __restrict__ tie_pieces, int* __restrict__ tie_counts, unsigned long long* __restrict__ tie_res)
{
    __shared__ __attribute__((aligned(1024))) int8_t tile[3 * AM * 128];     // three 8 KB tile buffers; the four 4 KB key dumps reuse them
    __shared__ uint32_t xtab[256];
    __shared__ uint32_t ttab[1024];
    assoc_body_fp4<false, 1>(q, nullptr, nq, mx, nullptr, nm_bound, nm_dev, nm_pad, m_chunk, max_distance, part, done, idx, dist, tile, nullptr, xtab, ttab, tie_pieces, tie_counts, tie_res);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_assoc_fp4_gated_s(const uint8_t* __restrict__ q, const uint8_t* __restrict__ qcolor, int nq,
                                               const int8_t* __restrict__ mx, const int8_t* __restrict__ mcx,
                                               int nm_bound, const int* __restrict__ nm_dev, int nm_pad, int m_chunk, int max_distance,
                                               unsigned int* __restrict__ part, int* __restrict__ done, int32_t* __restrict__ idx, float* __restrict__ dist,
                                               int* __restrict__ tie_pieces, int* __restrict__ tie_counts, unsigned long long* __restrict__ tie_res)
{
    __shared__ __attribute__((aligned(1024))) int8_t tile[3 * AM * 128];
    __shared__ __attribute__((aligned(1024))) int8_t ctile[3 * AM * 32];
    __shared__ uint32_t xtab[256];
    __shared__ uint32_t ttab[1024];
    assoc_body_fp4<true, 1>(q, qcolor, nq, mx, mcx, nm_bound, nm_dev, nm_pad, m_chunk, max_distance, part, done, idx, dist, tile, ctile, xtab, ttab, tie_pieces, tie_counts, tie_res);
}

// colour gated: one more matrix step per row block (gen_fp4 gated = True), the map's colour rows stream beside the tiles
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_assoc_fp4_gated(const uint8_t* __restrict__ q, const uint8_t* __restrict__ qcolor, int nq,
                                               const int8_t* __restrict__ mx, const int8_t* __restrict__ mcx,
                                               int nm_bound, const int* __restrict__ nm_dev, int nm_pad, int m_chunk, int max_distance,
                                               unsigned int* __restrict__ part, int* __restrict__ done, int32_t* __restrict__ idx, float* __restrict__ dist,
                                               int* __restrict__ tie_pieces, int* __restrict__ tie_counts, unsigned long long* __restrict__ tie_res)
{
    __shared__ __attribute__((aligned(1024))) int8_t tile[6 * AM * 128];
    __shared__ __attribute__((aligned(1024))) int8_t ctile[6 * AM * 32];
    __shared__ uint32_t xtab[256];
    __shared__ uint32_t ttab[1024];
    assoc_body_fp4<true>(q, qcolor, nq, mx, mcx, nm_bound, nm_dev, nm_pad, m_chunk, max_distance, part, done, idx, dist, tile, ctile, xtab, ttab, tie_pieces, tie_counts, tie_res);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_assoc(const uint8_t* __restrict__ q, const uint8_t* __restrict__ qcolor, int nq,
                                               const int8_t* __restrict__ mx, const int8_t* __restrict__ mcx,
                                               int nm_bound, const int* __restrict__ nm_dev, int nm_pad, int m_chunk, int max_distance,
                                               unsigned int* __restrict__ part, int* __restrict__ done, int32_t* __restrict__ idx, float* __restrict__ dist,
                                               int* __restrict__ tie_pieces, int* __restrict__ tie_counts, unsigned long long* __restrict__ tie_res)
{
    __shared__ __attribute__((aligned(1024))) int8_t tile[3 * AM * 256];     // triple buffered map tile
    __shared__ __attribute__((aligned(1024))) int8_t ctile[3 * AM * 32];     // the tiles' ninth-step operands
    __shared__ uint2 xtab[256];
    assoc_body<true>(q, qcolor, nq, mx, mcx, nm_bound, nm_dev, nm_pad, m_chunk, max_distance, part, done, idx, dist, tile, ctile, xtab, tie_pieces, tie_counts, tie_res);
}

// the same without colour gating: 8 MFMA steps per block (the block counter is the chain's start value), no ninth-step
// operands streamed
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_assoc_plain(const uint8_t* __restrict__ q, const uint8_t* __restrict__ qcolor, int nq,
                                               const int8_t* __restrict__ mx, const int8_t* __restrict__ mcx,
                                               int nm_bound, const int* __restrict__ nm_dev, int nm_pad, int m_chunk, int max_distance,
                                               unsigned int* __restrict__ part, int* __restrict__ done, int32_t* __restrict__ idx, float* __restrict__ dist,
                                               int* __restrict__ tie_pieces, int* __restrict__ tie_counts, unsigned long long* __restrict__ tie_res)
{
    __shared__ __attribute__((aligned(1024))) int8_t tile[3 * AM * 256];
    __shared__ uint2 xtab[256];
    assoc_body<false>(q, qcolor, nq, mx, mcx, nm_bound, nm_dev, nm_pad, m_chunk, max_distance, part, done, idx, dist, tile, nullptr, xtab, tie_pieces, tie_counts, tie_res);
}

__global__ void k_fill_u32(unsigned int* p, size_t n, unsigned int v)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void k_fill_nomatch(int n, int32_t* idx, float* dist)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { idx[i] = -1; dist[i] = -1.f; }
}

void launch_assoc_nomatch(int nq, int32_t* idx, float* dist, hipStream_t s)
{
    if (nq > 0) hipLaunchKernelGGL(k_fill_nomatch, dim3((nq + 255) / 256), dim3(256), 0, s, nq, idx, dist);
}

size_t assoc_rows_padded_m(int nm) { return ((size_t)nm + AM - 1) / AM * AM; }

// part[] (one word per query and map chunk, written before it is read by every launch) and done[] (one arrival counter
// per 256-query block, idle 0: the kernel leaves it idle, so it is cleared only when it is (re)allocated)
static hipError_t assoc_scratch_reserve(AssocScratch& w, size_t qblocks, size_t splits, size_t counters, hipStream_t s)
{
    hipError_t e;
    if (qblocks * splits > w.cap_part) {
        if (w.part) { if ((e = hipStreamSynchronize(s)) != hipSuccess) return e; (void)hipFree(w.part); w.part = nullptr; w.cap_part = 0; }
        const size_t cap = qblocks * splits + qblocks * splits / 2;
        if ((e = hipMalloc((void**)&w.part, cap * AQW * sizeof(unsigned int))) != hipSuccess) return e;
        w.cap_part = cap;
    }
    if (counters > w.cap_blocks) {                          // (one arrival counter per workgroup column: 128-query blocks in the small shape)
        if (w.done) { if ((e = hipStreamSynchronize(s)) != hipSuccess) return e; (void)hipFree(w.done); w.done = nullptr; w.cap_blocks = 0; }
        const size_t cap = counters + counters / 2 + 8;
        if ((e = hipMalloc((void**)&w.done, cap * sizeof(int))) != hipSuccess) return e;
        if ((e = hipMemsetAsync(w.done, 0, cap * sizeof(int), s)) != hipSuccess) return e;
        w.cap_blocks = cap;
    }
    return hipSuccess;
}

void assoc_scratch_free(AssocScratch& w)
{
    if (w.part) (void)hipFree(w.part);
    if (w.done) (void)hipFree(w.done);
    if (w.tie_list) (void)hipFree(w.tie_list);
    w.part = nullptr; w.done = nullptr; w.cap_part = 0; w.cap_blocks = 0; w.tie_list = nullptr; w.cap_list = 0;
}

void launch_assoc_pack_map(const uint8_t* codes, const uint8_t* colors, int n, int n_pad, int fp4, int8_t* x, int8_t* cx, hipStream_t s)
{
    if (n_pad <= 0) return;
    hipLaunchKernelGGL(k_assoc_pack_map, dim3(((size_t)n_pad * 32 + 255) / 256), dim3(256), 0, s, codes, colors, n, n_pad, fp4, x, cx);
}

// association of raw query codes against a packed map (the live map keeps its side packed across calls; lf_associate
// packs its caller's raw map first).  ONE launch.
hipError_t launch_assoc_core(const uint8_t* q, const uint8_t* qcolor, int nq, const int8_t* mx, const int8_t* mcx, int nm,
                       const int* nm_dev, int gating, int max_distance, AssocScratch& w, int32_t* idx, float* dist, hipStream_t s)
{
    const int nm_pad = (int)assoc_rows_padded_m(nm);
    const int tiles = nm_pad / AM;
    const int min_splits = (nm_pad + (kMaxBlocksPerChunk * 32) - 1) / (kMaxBlocksPerChunk * 32);
    // The shape: big = 256 queries per workgroup (two row blocks per wave, six tile buffers: 54 KB, 239 registers -- the fastest alone);
    // small = 128 (one row block, three buffers: 29 KB, ~160 registers -- the one that finds room beside other batches' region growing).
    // LF_ASSOC_SHAPE=big|small forces one; otherwise small up to kAssocSmallMax queries (the pipelined front end's batches), big beyond.
    static const int shape_env = getenv("LF_ASSOC_SHAPE") ? (getenv("LF_ASSOC_SHAPE")[0] == 's' ? 1 : (getenv("LF_ASSOC_SHAPE")[0] == 'b' ? 2 : 0)) : 0;
    static const bool force_i8 = getenv("LF_ASSOC_INT8") != nullptr;
    const bool small = !force_i8 && (shape_env == 1 || (shape_env == 0 && nq <= kAssocSmallMax));
    const int qw = small ? AQW / 2 : AQW;
    const int qblocks = (nq + qw - 1) / qw;
    // 2 workgroups are resident per CU: split the map so that the grid is one round of the 512 slots -- long chunks
    // amortise the query expansion and the final cross-lane reduction (measured: 512 > 1024 > 768 > 256); the small shape: 3 per CU
    static const int slots_env = getenv("LF_ASSOC_SLOTS") ? atoi(getenv("LF_ASSOC_SLOTS")) : 0;
    const int slots = slots_env > 0 ? slots_env : (small ? 1024 : 512);
    int splits = slots / qblocks;
    if (splits > 128) splits = 128;            // one or two query blocks: more, shorter chunks only lengthen the merge (32 -> 19 us at 256 x 50 000)
    if (splits < min_splits) splits = min_splits;
    if (splits > tiles) splits = tiles;
    if (splits < 1) splits = 1;
    const int m_chunk = (tiles + splits - 1) / splits * AM;
    splits = (nm_pad + m_chunk - 1) / m_chunk;
    const int qblocks256 = (nq + AQW - 1) / AQW;                   // what the scratch sizes and the tie pass count in
    hipError_t e = assoc_scratch_reserve(w, (size_t)qblocks256, (size_t)splits, (size_t)qblocks, s);
    if (e != hipSuccess) return e;
    w.qblocks = qblocks256; w.splits = splits; w.m_chunk = m_chunk;
    // the tie pass's lists are written by this launch's merge step when the caller asked for them (w.tie_res: the result words)
    int *tie_pieces = nullptr, *tie_counts = nullptr;
    if (w.tie_res) {
        const size_t n_pieces = (size_t)qblocks256 * 4;
        const size_t need = (size_t)splits * n_pieces * 64 + (size_t)splits * n_pieces;
        if (need > w.cap_list) {
            if (w.tie_list) (void)hipFree(w.tie_list);
            w.tie_list = nullptr; w.cap_list = 0;
            e = hipMalloc(&w.tie_list, (need + need / 2) * sizeof(int));
            if (e != hipSuccess) return e;
            w.cap_list = need + need / 2;
        }
        tie_pieces = w.tie_list;
        tie_counts = w.tie_list + (size_t)splits * n_pieces * 64;
    }
    unsigned long long* tie_res = w.tie_res;
    // ungated: the FP4 kernel (its map operands are e2m1 rows: MapDevice::fp4 / launch_assoc_pack_map(fp4 = 1));
    // LF_ASSOC_INT8=1 keeps the int8 kernel for A/B runs -- the caller's operands must then be int8 rows
    if (small && gating) hipLaunchKernelGGL(k_assoc_fp4_gated_s, dim3(qblocks, splits), dim3(256), 0, s, q, qcolor, nq, mx, mcx, nm, nm_dev, nm_pad, m_chunk, max_distance, w.part, w.done, idx, dist, tie_pieces, tie_counts, tie_res);
    else if (small) hipLaunchKernelGGL(k_assoc_fp4_s, dim3(qblocks, splits), dim3(256), 0, s, q, nq, mx, nm, nm_dev, nm_pad, m_chunk, max_distance, w.part, w.done, idx, dist, tie_pieces, tie_counts, tie_res);
    else if (gating && !force_i8) hipLaunchKernelGGL(k_assoc_fp4_gated, dim3(qblocks, splits), dim3(256), 0, s, q, qcolor, nq, mx, mcx, nm, nm_dev, nm_pad, m_chunk, max_distance, w.part, w.done, idx, dist, tie_pieces, tie_counts, tie_res);
    else if (gating) hipLaunchKernelGGL(k_assoc, dim3(qblocks, splits), dim3(256), 0, s, q, qcolor, nq, mx, mcx, nm, nm_dev, nm_pad, m_chunk, max_distance, w.part, w.done, idx, dist, tie_pieces, tie_counts, tie_res);
    else if (!force_i8) hipLaunchKernelGGL(k_assoc_fp4, dim3(qblocks, splits), dim3(256), 0, s, q, nq, mx, nm, nm_dev, nm_pad, m_chunk, max_distance, w.part, w.done, idx, dist, tie_pieces, tie_counts, tie_res);
    else hipLaunchKernelGGL(k_assoc_plain, dim3(qblocks, splits), dim3(256), 0, s, q, qcolor, nq, mx, mcx, nm, nm_dev, nm_pad, m_chunk, max_distance, w.part, w.done, idx, dist, tie_pieces, tie_counts, tie_res);
#ifdef LF_ASSOC_STAMPS
    {
        static int calls = 0;
        if (++calls == 12) {
            (void)hipDeviceSynchronize();
            static unsigned long long h[8 * 8192];
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_assoc_stamps), sizeof(h));
            const int n = qblocks * splits < 8192 ? qblocks * splits : 8192;
            unsigned long long w0 = ~0ull, w1 = 0; double d[8] = { 0 };
            double e6 = 0, e7 = 0;
            for (int i = 0; i < n; ++i) { const unsigned long long* t = h + 8 * i; if (t[0] < w0) w0 = t[0]; if (t[0] > w1) w1 = t[0]; for (int k = 2; k <= 5; ++k) d[k] += (double)(t[k] - t[k - 1]); e6 += (double)(t[6] - t[1]); e7 += (double)(t[7] - t[1]); }
            fprintf(stderr, "[assoc stamps] tables + barrier at +%.0f, fragments expanded at +%.0f cycles\n", e6 / n, e7 / n);
            fprintf(stderr, "[assoc stamps] %d workgroups (%d x %d): start skew %.2f us | setup %.0f  loop %.0f  reduce %.0f  publish %.0f cycles (mean per workgroup)\n",
                    n, qblocks, splits, (double)(w1 - w0) / 100.0, d[2] / n, d[3] / n, d[4] / n, d[5] / n);
        }
    }
#endif
    return hipGetLastError();
}

hipError_t launch_assoc(const uint8_t* q, int nq, const uint8_t* m, int nm, int8_t* mx, int8_t* mcx, AssocScratch& w, int32_t* idx,
                        float* dist, hipStream_t s)
{
    static const bool force_i8 = getenv("LF_ASSOC_INT8") != nullptr;
    launch_assoc_pack_map(m, nullptr, nm, (int)assoc_rows_padded_m(nm), force_i8 ? 0 : 1, mx, mcx, s);
    return launch_assoc_core(q, nullptr, nq, mx, mcx, nm, nullptr, 0, 128, w, idx, dist, s);
}

// ---------------------------------------------------------------- float LBD (72-d)
// dist^2 = |q|^2 + |m|^2 - 2 q.m ; the dot product runs on the fp32-input MFMA (K = 2 per
// instruction, 36 steps).  One wave = 32 queries x 32 map entries per step.
__global__ void k_sqnorm72(const float* __restrict__ x, int n, float* __restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0;
    for (int k = 0; k < 72; ++k) { float v = x[(size_t)i * 72 + k]; s += v * v; }
    out[i] = s;
}

__global__ __launch_bounds__(256) void k_assoc_float(const float* __restrict__ q, const float* __restrict__ qn, int nq,
                                                     const float* __restrict__ m, const float* __restrict__ mn, int nm,
                                                     int m_chunk, unsigned long long* __restrict__ best)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    if (q0 >= nq) return;
    const int r32 = lane & 31, half = lane >> 5;
    const int qi = min(q0 + r32, nq - 1);
    float A[36];
#pragma unroll
    for (int s = 0; s < 36; ++s) A[s] = q[(size_t)qi * 72 + 2 * s + half];
    float qnr[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        qnr[r] = qn[min(q0 + row, nq - 1)];
    }
    unsigned long long running[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) running[r] = ~0ull;
    const int m_begin = blockIdx.y * m_chunk, m_end = min(nm, m_begin + m_chunk);
    for (int m0 = m_begin; m0 < m_end; m0 += 32) {
        const int col = m0 + r32;
        const int ci = min(col, nm - 1);
        v16f acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 36; ++s) {
            float B = m[(size_t)ci * 72 + 2 * s + half];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[s], B, acc, 0, 0, 0);
        }
        const float mnc = mn[ci];
        const bool valid = col < m_end;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float d2 = (qnr[r] + mnc) - 2.f * acc[r];
            d2 = d2 < 0.f ? 0.f : d2;
            unsigned long long v = valid ? (((unsigned long long)__float_as_uint(d2) << 32) | (unsigned int)col) : ~0ull;
            running[r] = v < running[r] ? v : running[r];
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        unsigned long long v = running[r];
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) {
            unsigned long long o = __shfl_xor(v, d);
            v = o < v ? o : v;
        }
        if (r32 == 0) {
            int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            int qq = q0 + row;
            if (qq < nq) atomicMin(best + qq, v);
        }
    }
}

__global__ void k_assoc_float_finish(const unsigned long long* __restrict__ best, int nq, int32_t* __restrict__ idx,
                                     float* __restrict__ dist)
{
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    unsigned long long v = best[q];
    if (v == ~0ull) { idx[q] = -1; dist[q] = -1.f; }
    else { idx[q] = (int)(v & 0xffffffffull); dist[q] = dm::fsqrt(__uint_as_float((unsigned int)(v >> 32))); }
}

__global__ void k_fill_u64(unsigned long long* p, int n, unsigned long long v)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

void launch_assoc_float(const float* q, int nq, const float* m, int nm, float* qn, float* mn,
                        unsigned long long* best, int32_t* idx, float* dist, hipStream_t s)
{
    hipLaunchKernelGGL(k_sqnorm72, dim3((nq + 255) / 256), dim3(256), 0, s, q, nq, qn);
    hipLaunchKernelGGL(k_sqnorm72, dim3((nm + 255) / 256), dim3(256), 0, s, m, nm, mn);
    hipLaunchKernelGGL(k_fill_u64, dim3((nq + 255) / 256), dim3(256), 0, s, best, nq, ~0ull);
    const int qblocks = (nq + 127) / 128;
    int splits = (1024 + qblocks - 1) / qblocks;
    const int tiles = (nm + 31) / 32;
    if (splits > tiles) splits = tiles;
    if (splits < 1) splits = 1;
    const int m_chunk = (tiles + splits - 1) / splits * 32;
    splits = (nm + m_chunk - 1) / m_chunk;
    hipLaunchKernelGGL(k_assoc_float, dim3(qblocks, splits), dim3(256), 0, s, q, qn, nq, m, mn, nm, m_chunk, best);
    hipLaunchKernelGGL(k_assoc_float_finish, dim3((nq + 255) / 256), dim3(256), 0, s, best, nq, idx, dist);
}

}  // namespace lf
